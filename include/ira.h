/*
 * ira.h -- C-ABI of libira.so: batched impulse-response analysis kernels for AMD MI355X (gfx950).
 *
 * The reference project (kianmcevoy/audio_analysis) is pure Python and has no FFI for this path
 * (SURVEY.md section 8b); the boundary it exposes is its per-module Python API.  Each entry point
 * below therefore cites the reference Python function whose numeric body it replaces, and
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no exceptions, no torch types.
 *  - Every pointer named *_dev is DEVICE memory owned by the caller (the Python host allocates
 *    torch-ROCm tensors and passes data_ptr()).  The library allocates nothing the caller sees and
 *    keeps no state: all tables (windows, twiddles) are caller-provided device buffers.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls only enqueue work;
 *    they never synchronise.  Re-entrant per stream.
 *  - Ragged batches: a flat float32 sample buffer plus per-segment int64 offset/length arrays.
 *  - Return value: 0 = ok; negative = argument error (IRA_E_*); <= -1000 = -(1000 + hipError_t).
 */
#ifndef IRA_H_
#define IRA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRA_OK 0
#define IRA_E_NULL (-1)       /* a required pointer was NULL */
#define IRA_E_SIZE (-2)       /* a size/shape argument is out of the supported range */
#define IRA_E_UNSUPPORTED (-3)
#define IRA_E_IO (-4)         /* a file could not be opened or read (host-side ingest entry points only) */
#define IRA_E_FORMAT (-5)     /* a file is not RIFF/WAVE (host-side ingest entry points only) */
#define IRA_E_HIP_BASE (-1000)

#define IRA_ABI_VERSION 8   /* bumped whenever an exported signature or a scratch-size constant changes */

int32_t ira_abi_version(void);
const char* ira_error_string(int32_t code);

/* ---- a2: peak pick ---------------------------------------------------------------------------
 * peak_dev[s] = argmax_n |x[off[s]+n]|, n < len[s]; the FIRST maximum wins (bit-exact integer).
 * Replaces np.argmax(np.abs(x)) at reference analyse/decay.py:136, spectrogram.py:181,
 * waterfall.py:359, modalcloud.py:299, frequency_response.py:186, filterplot.py:125,
 * zplane.py:197, rt60bands.py:335.  Also returns the peak magnitude (zplane.py:211).
 * max_len = longest segment (sizes the grid; every sample of every segment is visited; < 2^32); nseg <= 65535. */
int32_t ira_peak_index(const float* x_dev, const int64_t* off_dev, const int64_t* len_dev,
                       int32_t nseg, int64_t max_len, int64_t* peak_dev, float* peak_abs_dev, void* stream);

/* ---- a3: Schroeder energy-decay curve ----------------------------------------------------------
 * For each segment: e = x^2 (f64) -> reverse cumulative sum -> max(.,eps) -> /edc[0] -> 10 log10
 * -> max(.,floor_db) -> float32, written to edc_db_dev at the same offsets as the input segment
 * (edc_off_dev).  Replaces compute_schroeder_edc_db, reference analyse/decay.py:115-170
 * edc_db64_dev (optional, may be NULL): the same curve as float64 BEFORE the floor, for the optional
 * host-side dB smoothing of decay.py:161-164.  edc_db_dev may be NULL if only that is wanted.
 * max_len = longest segment (sizes the grid; up to 2047 tiles of 4096 samples = 8.38 M samples).
 * scratch_dev: nseg * IRA_EDC_SCRATCH_DOUBLES doubles. */
#define IRA_EDC_SCRATCH_DOUBLES 4096
int32_t ira_edc_db(const float* x_dev, const int64_t* off_dev, const int64_t* len_dev, int32_t nseg,
                   int64_t max_len, double eps, double floor_db, float* edc_db_dev, double* edc_db64_dev,
                   const int64_t* edc_off_dev, double* scratch_dev, void* stream);

/* ---- a4/a5/a16: threshold crossings + straight-line decay fits on a float32 dB curve -------------
 * For each curve c (ncurves of them) and each of nranges (hi_db, lo_db_effective) pairs:
 *   first index with y <= target (float32 compare), linear interpolation of the crossing time in
 *   f64, mask t>=t_start & t<=t_end in float32, >= min_points, least-squares line, slope<0, r^2,
 *   rt60 = -60/slope.
 * Replaces _interpolated_crossing_time_seconds + fit_decay_slope_over_db_range, reference
 * analyse/decay.py:173-260, and the per-bin copy analyse/modalcloud.py:215-281.
 * Time axis: t[i] = float32(i) * t_mul / t_div evaluated in float32 exactly like the reference
 * (decay.py:169: t_mul = 1, t_div = sr; spectrogram.py:158: t_mul = hop, t_div = sr).
 * If t_axis_dev != NULL it is an explicit float32 time axis (>= max_len values, shared by all curves)
 * used instead of the analytic one (the public fit_decay_slope_over_db_range takes any time array).
 * If rel_to_peak != 0 the curve is first shifted by its own maximum in float32
 * (modalcloud.py:356-361) and curves that are not finite or whose peak - floor_db <
 * min_peak_above_floor are flagged invalid (status bit 2).
 * ranges_hi_lo (2*nranges doubles) and cross_targets are HOST arrays; nranges <= 4, ncross <= 4;
 * max_len (longest curve) only picks the workgroup size.
 * cross_targets (ncross values) -> cross_out_dev[c*ncross + j] crossing time or NaN
 * (used for the 0 / -10 dB early-decay time, decay.py:280-286).
 * fit_out_dev: ncurves*nranges records of IRA_FIT_DOUBLES doubles:
 *   [0] valid (1/0) [1] start_t [2] end_t [3] slope [4] intercept [5] r2 [6] rt60 [7] npts */
#define IRA_FIT_DOUBLES 8
int32_t ira_curve_fits(const float* y_dev, const int64_t* off_dev, const int64_t* len_dev,
                       int32_t ncurves, int32_t max_len, float t_mul, float t_div,
                       const float* t_axis_dev, const double* ranges_hi_lo,
                       int32_t nranges, int32_t min_points, const double* cross_targets,
                       int32_t ncross, int32_t rel_to_peak, double floor_db,
                       double min_peak_above_floor, double* fit_out_dev, double* cross_out_dev,
                       void* stream);

/* Optional dB smoothing of the EDC (reference analyse/decay.py:159-166, default off): out[s][i] = float32(max(floor_db,
 * numpy.convolve(edc_db64[s], ones(window)/window, mode="same")[i])) on the unfloored float64 curve ira_edc_db writes
 * (edc_db64_dev); in and out share the offsets off_dev.  IRA_E_UNSUPPORTED when a segment is shorter than the window
 * (numpy's "same" then changes the length of the curve). */
int32_t ira_edc_box_smooth(const double* edc_db64_dev, const int64_t* off_dev, const int64_t* len_dev, int32_t nseg,
                           int64_t max_len, int32_t window, double floor_db, float* out_dev, void* stream);

/* ---- a3-a6 fused: Schroeder EDC -> crossings -> decay-line fits, straight from the samples --------------------------
 * The same results as ira_edc_db followed by ira_curve_fits (analytic time axis t[i] = float32(i)*t_mul/t_div, no
 * rel_to_peak), without reading an EDC array: per segment the chunk sums locate each target level, only the
 * chunks around the crossings and between them are re-scanned (identical float32 dB values, same code as the emit
 * pass), and the regression is one sweep of shifted float64 moments.  Replaces, for one segment, the call chain
 * compute_schroeder_edc_db -> _interpolated_crossing_time_seconds -> fit_decay_slope_over_db_range of
 * analyse_decay_for_channel (reference analyse/decay.py:268-329) and of every band of
 * _compute_band_metrics_from_samples (analyse/rt60bands.py:272-321), whose EDC curve is never a result.
 * edc_db_dev (optional, may be NULL): also write the float32 dB curve at edc_off_dev[s] (the decay block returns it).
 * fit_out_dev / cross_out_dev: records exactly as ira_curve_fits writes them.  ranges_hi_lo / cross_targets: HOST arrays.
 * scratch_dev: nseg * IRA_EDC_SCRATCH_DOUBLES doubles.  max_len as for ira_edc_db.
 * tile_part_dev (optional, may be NULL; round 5): partial tile energies a producer of the segments left behind
 * (ira_band_irfft_smooth, tile_part_dev): segment s with part_off_dev[s] >= 0 takes the energy of its tile j (4096 samples,
 * counted from the END of the segment) as the sum over w < part_wgs_dev[s] of tile_part_dev[part_off_dev[s] +
 * w * part_tiles_dev[s] + j] (part_tiles_dev[s] = tiles of the producer's whole signal), for every tile but the one that holds its first sample (that one is scanned, so that edc[0] stays
 * the value the curve has at index 0) -- instead of reading its samples once more (rt60bands.py:272-321 through
 * decay.py:115-170: a band's signal is written by the inverse transform a moment earlier).  CONTRACT: such a segment ends
 * where the producer's signal ends (a band signal from its start index on).  part_off_dev[s] < 0: the samples are read. */
int32_t ira_edc_fits(const float* x_dev, const int64_t* off_dev, const int64_t* len_dev, int32_t nseg,
                     int64_t max_len, double eps, double floor_db, float t_mul, float t_div,
                     const double* ranges_hi_lo, int32_t nranges, int32_t min_points,
                     const double* cross_targets, int32_t ncross, double* fit_out_dev, double* cross_out_dev,
                     float* edc_db_dev, const int64_t* edc_off_dev, double* scratch_dev, const double* tile_part_dev,
                     const int64_t* part_off_dev, const int32_t* part_wgs_dev, const int32_t* part_tiles_dev,
                     void* stream);

/* ---- a11: STFT magnitude in dB -------------------------------------------------------------------
 * Valid framing (no padding), frame f of segment s starts at off[s] + f*hop, nframes[s] frames.
 * out[s] is a C-contiguous (n_fft/2+1, nframes[s]) float32 matrix at out_dev + out_off[s]:
 *   20*log10(max(|rfft(frame*window)|, 10^(floor_db/20))).
 * Replaces _compute_stft_magnitude_db, reference analyse/spectrogram.py:107-160 and its copies
 * analyse/waterfall.py:188-230, analyse/modalcloud.py:121-158.
 * precision: 32 = float32 butterflies (spectrogram), 64 = float64 butterflies.
 * window_dev: n_fft values; twiddle_dev: n_fft/2 interleaved (cos, -sin) pairs of
 * exp(-2*pi*i*k/n_fft); both float (precision 32) or double (precision 64).
 * frame_sel_dev (optional): if not NULL, nframes[s] is the number of SELECTED frames and
 * frame_sel_dev[sel_off[s] + j] gives the frame index of output column j (waterfall shortcut,
 * reference analyse/waterfall.py:309). n_fft: power of two, 64..16384. */
int32_t ira_stft_mag_db(const float* x_dev, const int64_t* off_dev, const int32_t* nframes_dev,
                        int32_t nseg, int32_t max_frames, int32_t n_fft, int32_t hop,
                        const void* window_dev, const void* twiddle_dev, int32_t precision,
                        double floor_db, float* out_dev, const int64_t* out_off_dev,
                        const int32_t* frame_sel_dev, const int64_t* sel_off_dev, void* stream);

/* Same transform, FRAME-MAJOR output: out[e] is a (T_e, n_fft/2+1) matrix, i.e. the transpose of the reference's
 * (F, T) array -- every frame's bins are contiguous, which lets each wave store its frame directly (no transposing
 * tile, no partial-line writes).  For device-resident consumers and hosts that take a `.T` view.  Implemented for
 * precision 32 / n_fft 4096 and precision 64 / n_fft 8192 (the reference's spectrogram and modal-cloud defaults);
 * anything else returns IRA_E_UNSUPPORTED (use ira_stft_mag_db). */
int32_t ira_stft_mag_db_tf(const float* x_dev, const int64_t* off_dev, const int32_t* nframes_dev,
                           int32_t nseg, int32_t max_frames, int32_t n_fft, int32_t hop,
                           const void* window_dev, const void* twiddle_dev, int32_t precision,
                           double floor_db, float* out_dev, const int64_t* out_off_dev,
                           const int32_t* frame_sel_dev, const int64_t* sel_off_dev, void* stream);

/* STFT + modal-cloud log-bin aggregation in ONE kernel (ira_stft_mag_db followed by ira_logbin_aggregate, with the dB
 * matrix never written): curves_dev[e] is the (nbins, T_e) float32 matrix at curves_off_dev[e]; k_base / first / count
 * as in ira_logbin_aggregate (every row k_base + first[b] + j must be < n_fft/2 + 1).  Same arithmetic, including the
 * float32 rounding of the dB value before it is converted back to linear magnitude.  precision 64 / n_fft 8192 only
 * (IRA_E_UNSUPPORTED otherwise).  Replaces reference analyse/modalcloud.py:121-158 + :176-207 for the modal cloud. */
int32_t ira_stft_logbin(const float* x_dev, const int64_t* off_dev, const int32_t* nframes_dev, int32_t nseg,
                        int32_t max_frames, int32_t n_fft, int32_t hop, const void* window_dev,
                        const void* twiddle_dev, int32_t precision, double floor_db, int32_t k_base,
                        const int32_t* first_dev, const int32_t* count_dev, int32_t nbins, float* curves_dev,
                        const int64_t* curves_off_dev, void* stream);

/* ---- a9/a17: arbitrary-length float64 DFTs (Bluestein over a four-step FFT of size M) -------------------
 * Common arguments: the convolution size m = M, a power of two (16 <= M <= 2^22) or three times one
 * (96 <= M <= 3*2^20), with M >= 2*max(L) - 1 -- or only M >= L + L/2 for ira_rfft_any elements that carry ONE real
 * signal (no x2off / interleave): their L/2+1 bins do not see the wrap-around; three caller-provided
 * complex-f64 tables for M = N1*N2 with the split ira_fft_split() reports (N2 a power of two, N1 one or 3x one):
 *   t1_dev[k] = exp(-2 pi i k/N1), k < N1;  t2_dev[k] = exp(-2 pi i k/N2), k < N2;
 *   tf_dev[k] = exp(-2 pi i k/M),  k < N2.
 * work_dev: nb * M complex f64 of scratch.  bfilt_dev: chirp-filter spectra built by ira_bluestein_filter
 * FOR THE SAME M, M complex f64 each; bidx_dev[e] selects the filter (the one built for length L[e]) of element e. */

/* The four-step split the library uses for size m (host code sizes t1/t2/tf from it); IRA_E_SIZE if m is not a
 * supported size.  No reference counterpart: numpy's pocketfft plans internally (reference
 * analyse/frequency_response.py:204). */
int32_t ira_fft_split(int32_t m, int32_t* n1, int32_t* n2);

/* bfilt_dev[j] = FFT_M of the Bluestein chirp filter for length L_dev[j], j < nfilt. */
int32_t ira_bluestein_filter(const int32_t* L_dev, int32_t nfilt, int32_t m, const void* t1_dev,
                             const void* t2_dev, const void* tf_dev, double* bfilt_dev, void* stream);

/* spec_out[e][k] = sum_n x[xoff[e]+n] * (hanning(L[e])[n] | 1) * exp(-2 pi i n k / L[e]),  k = 0..L[e]/2,
 * complex f64 at spec_out_dev + 2*spec_off_dev[e] doubles.  Replaces np.fft.rfft(x * w) of arbitrary
 * length at reference analyse/frequency_response.py:204-213, analyse/filterplot.py:145-152 and the
 * forward transform of analyse/rt60bands.py:172.
 * Two real signals of the SAME length can share one transform (z = x1 + i*x2, split by Hermitian symmetry):
 * x2off_dev (may be NULL = no pairs) gives the second signal of element e or -1; its spectrum goes to
 * spec_off2_dev[e]; zpair_dev/zpair_off_dev is scratch for the full-length complex DFT (L[e] complex f64 per
 * paired element) and max_len >= max L[e].  Cross-talk between the two signals is at the 1e-16 level of the
 * larger one.
 * Zero-padded / truncated transforms, numpy.fft.rfft(x * hanning(len(x)), n=L) (reference
 * analyse/group_delay.py:95-109): data_len_dev[e] (may be NULL = L[e]) samples are read, the rest of the L[e]
 * inputs are zero (data_len > L truncates), and the Hann window is the one of length win_len_dev[e] (NULL =
 * L[e]); data_len2_dev / win_len2_dev (NULL = same as the first) apply to the second signal of a pair.
 * interleave_dev (may be NULL): interleave[e] = 1 makes element e ONE real signal of even length 2*L[e] carried as the
 * complex sequence x[2m] + i*x[2m+1] (x2off_dev[e] must be xoff_dev[e] + 1, zpair scratch L[e] complex): half the
 * transform size; its L[e]+1 spectrum bins go to spec_off_dev[e].  Window lengths then refer to the real signal
 * (win_len = 2*L[e]).
 * keep_packed != 0 (round 4): the L[e]-point transforms Z of interleaved elements are NOT untangled -- they stay where
 * the last pass wrote them, zpair_dev + 2*zpair_off_dev[e] (the caller may point that into spec_out_dev: L[e] values fit
 * the element's L[e]+1 bins), and the consumer forms X[k] = E[k] + W^k O[k] as it reads them
 * (ira_spectrum_mag_phase, packed_dev): one read-modify-write pass over every such spectrum less.  Such a call must not
 * carry paired elements (x2off_dev[e] >= 0 only where interleave_dev[e] = 1): no split kernel is launched at all.
 * keep_packed with x2off_dev but WITHOUT interleave_dev (every second signal would be a paired one) returns
 * IRA_E_UNSUPPORTED. */
int32_t ira_rfft_any(const float* x_dev, const int64_t* xoff_dev, const int32_t* L_dev, int32_t nb,
                     int32_t use_hann, int32_t m, const void* t1_dev, const void* t2_dev,
                     const void* tf_dev, const double* bfilt_dev, const int32_t* bidx_dev,
                     double* work_dev, double* spec_out_dev, const int64_t* spec_off_dev,
                     const int64_t* x2off_dev, const int64_t* spec_off2_dev, double* zpair_dev,
                     const int64_t* zpair_off_dev, int32_t max_len, const int32_t* data_len_dev,
                     const int32_t* win_len_dev, const int32_t* data_len2_dev, const int32_t* win_len2_dev,
                     const int32_t* interleave_dev, int32_t keep_packed, void* stream);

/* Band filter bank: element e takes the half spectrum at spec_dev + 2*spec_off_dev[e] (length L[e]/2+1),
 * multiplies it by TWO real masks (band_params_dev: 2 records of IRA_BAND_DOUBLES doubles per element:
 * [0] kind 0 none/1 low-pass/2 high-pass/3 band-pass, [1] hp ramp start, [2] hp ramp end (pass edge),
 * [3] lp ramp start (pass edge), [4] lp ramp end; masks are evaluated in float32 on the float32 axis
 * float32(k*freq_val[e]) exactly as reference analyse/rt60bands.py:116-167) and inverse-transforms both
 * at once (y1 + i*y2), writing float32 signals of length L[e] at y_dev + y1_off[e] and y_dev + y2_off[e]
 * (y2_off[e] < 0: no second band).  spec_off2_dev (may be NULL) lets the SECOND band come from another
 * spectrum of the same length and bin step (e.g. the odd band of two different channels share one inverse
 * transform): band 2 of element e then filters spec_dev + 2*spec_off2_dev[e].  Replaces _apply_fft_mask, reference analyse/rt60bands.py:170-175. */
#define IRA_BAND_DOUBLES 8
int32_t ira_band_irfft(const double* spec_dev, const int64_t* spec_off_dev, const int32_t* L_dev,
                       int32_t nb, const double* band_params_dev, const double* freq_val_dev,
                       int32_t m, const void* t1_dev, const void* t2_dev, const void* tf_dev,
                       const double* bfilt_dev, const int32_t* bidx_dev, double* work_dev, float* y_dev,
                       const int64_t* y1_off_dev, const int64_t* y2_off_dev,
                       const int64_t* spec_off2_dev, void* stream);

/* ---- a17/a18: spectrum post-processing -------------------------------------------------------------------
 * mag_db[e][k] = float32(20 log10(max(|X|, 10^(floor_db/20)))), optional phase[e][k] = atan2(im, re) (f64).
 * Reference analyse/frequency_response.py:213-218, analyse/filterplot.py:152-160.
 * packed_dev (may be NULL): packed[e] != 0 says element e (even L[e]) is still the PACKED half-length transform
 * Z = DFT(x[2m] + i x[2m+1]) of L[e]/2 values (ira_rfft_any, keep_packed) at spec_off_dev[e]; its L[e]/2+1 bins are formed
 * on the fly. */
int32_t ira_spectrum_mag_phase(const double* spec_dev, const int64_t* spec_off_dev, const int32_t* L_dev,
                               int32_t nb, int32_t max_len, double floor_db, float* mag_db_dev,
                               const int64_t* mag_off_dev, double* phase_dev,
                               const int64_t* phase_off_dev, const int32_t* packed_dev, void* stream);

/* numpy.unwrap (optional) + rad2deg (optional) -> float32 (out_dev, may be NULL), and/or the unwrapped phase in
 * float64 radians (out64_dev, may be NULL; same offsets).  Reference analyse/filterplot.py:162-168 and
 * analyse/group_delay.py:113-115. */
int32_t ira_phase_unwrap(const double* phase_dev, const int64_t* phase_off_dev, const int32_t* L_dev,
                         int32_t nb, int32_t do_unwrap, int32_t to_degrees, float* out_dev,
                         const int64_t* out_off_dev, double* out64_dev, void* stream);

/* Section 8f, group delay: gd[e][k] = -numpy.gradient(phase[e], w)[k] with w[k] = 2 pi ((k * bin_step[e]) /
 * sample_rate) in rad/sample, k < nbins[e]; numpy's rule "uniform formula only if every diff(w) is bit-identical,
 * three-point non-uniform formula otherwise" is reproduced (flags_dev: nb int32 of scratch, 1 = non-uniform).
 * phase/gd share off_dev (float64 elements).  Replaces the gradient step of _compute_group_delay_from_ir,
 * reference analyse/group_delay.py:117-124.   flags_known != 0 (round 4): flags_dev already holds, per element, whether numpy.gradient takes the
 * non-uniform formula (a function of bins, bin step and sample rate alone: callers decide it once per transform length);
 * the sweep that works it out on the device is skipped. */
int32_t ira_group_delay(const double* phase_dev, const int64_t* off_dev, const int32_t* nbins_dev, int32_t nb,
                        int32_t max_bins, const double* bin_step_dev, double sample_rate_hz, int32_t* flags_dev,
                        int32_t flags_known, double* gd_dev, void* stream);

/* Statistics over bins with f_min <= float32(k*freq_val[e]) <= f_max (float32 compares): out_dev[e*8..]:
 * [0] bin count [1] argmax bin of mag_db (first max) [2] its frequency [3] sum f*10^(dB/20) [4] sum 10^(dB/20)
 * [5] first in-range frequency [6] argmin |f - probe_hz| over ALL bins (first min) [7] mag_db there.
 * Reference analyse/frequency_response.py:238-260, analyse/filterplot.py:173-191. */
int32_t ira_spectrum_stats(const float* mag_db_dev, const int64_t* mag_off_dev, const int32_t* L_dev,
                           int32_t nb, const double* freq_val_dev, double f_min_hz, double f_max_hz,
                           double probe_hz, double* out_dev, void* stream);

/* ---- a14: waterfall slice normalisation ----------------------------------------------------------------
 * mag[e] is the (F, S_e) STFT dB matrix of the S_e selected frames (ira_stft_mag_db with frame_sel_dev);
 * out[e] is (S_e, nsel) float32: clip(mag[k_lo+k, s] - ref, -dyn_db, 0), ref = max over the selected block
 * (slice_max = 0) or over each slice (slice_max = 1), float32 arithmetic.
 * Replaces _build_rel_db_slices, reference analyse/waterfall.py:289-341 (smoothing off). */
int32_t ira_waterfall_rel(const float* mag_dev, const int64_t* mag_off_dev, const int32_t* nslices_dev,
                          int32_t nb, int32_t k_lo, int32_t nsel, int32_t slice_max, double dyn_db,
                          float* out_dev, const int64_t* out_off_dev, void* stream);

/* ---- a15: modal-cloud log-frequency aggregation -----------------------------------------------------------
 * mag[e] is the (F, T_e) STFT dB matrix; for log bin b rows k_base+first[b] .. +count[b]-1 are averaged as
 * linear magnitude 10^(dB/20) in float64 (rows added in order), then 20 log10(max(mean, 1e-30)) -> float32;
 * count[b] == 0 gives a NaN row.  out[e] is (nbins, T_e).  frame_major_rows > 0: mag[e] is the FRAME-MAJOR (T_e, F)
 * matrix of ira_stft_mag_db_tf with F = frame_major_rows (same results, same row order).
 * Replaces _aggregate_to_log_bins, reference analyse/modalcloud.py:176-207. */
int32_t ira_logbin_aggregate(const float* mag_dev, const int64_t* mag_off_dev, const int32_t* nframes_dev,
                             int32_t nb, int32_t max_frames, int32_t k_base, const int32_t* first_dev,
                             const int32_t* count_dev, int32_t nbins, float* out_dev,
                             const int64_t* out_off_dev, int32_t frame_major_rows, void* stream);

/* Optional log-frequency smoothing of dB curves IN PLACE (reference analyse/waterfall.py:140-185 and
 * analyse/frequency_response.py:117-169; default off): curve c = the bins k_lo[c] .. k_lo[c]+nsel[c]-1 (frequencies
 * float32(k * fstep[c])) of the float32 array at mag_dev + off[c], consecutive bins stride[c] elements apart.  They are
 * interpolated (numpy.interp) onto count[c] points uniform in log2(f) between log2_lo[c] and log2_hi[c] (the host computes
 * numpy.log2 of the first / last selected frequency and the count), averaged with a `window`-point box
 * (numpy.convolve(.., "same")), interpolated back and cast to float32.  through_float32 != 0: the waterfall variant, which
 * casts the gridded curve to float32 before and after the convolution.  max_count = largest count (<= 2048, else
 * IRA_E_UNSUPPORTED). */
int32_t ira_log_smooth_db(float* mag_dev, const int64_t* off_dev, const int32_t* stride_dev, const int32_t* k_lo_dev,
                          const int32_t* nsel_dev, const double* fstep_dev, const double* log2_lo_dev,
                          const double* log2_hi_dev, const int32_t* count_dev, int32_t ncurves, int32_t max_count,
                          int32_t window, int32_t through_float32, void* stream);

/* ---- a19-a21: z-plane AR pole fit ----------------------------------------------------------------------------
 * Covariance-method AR least squares of order `order` on segments x[xoff[e] .. +len[e]) / divisor[e]
 * (divisor_dev may be NULL = 1; if x64_dev != NULL the samples are read from it as float64 instead): normal equations G = A^T A, r = A^T y with A[n,k] = s[n-k], y = -s[n],
 * n = order..len-1, contracted in float64 on the matrix cores, optional ridge added to the diagonal, Cholesky
 * solve.  coeffs_dev[e*(order+1) ..] = [1, a_1 .. a_order].  Replaces _fit_ar_least_squares, reference
 * analyse/zplane.py:83-120 (which uses an SVD-based lstsq; results agree to ~cond(A)^2 * 1e-16).
 * partial_dev: nb * ira_ar_partial_doubles(order, max_len) doubles of scratch; gscratch_dev: nb*order*order
 * doubles, only needed when order > 128; info_dev (optional): IRA_AR_INFO_DOUBLES doubles per element
 * [0] status: 0 solved, 1 a Cholesky pivot was not positive, 2 solved and refined (ira_ar_refine), 3 not finite,
 * 4 rank-deficient: minimum-norm solution (ira_ar_minnorm, which then redefines [1..3]), 5 solved by the double-double
 * normal equations (ira_ar_exact), [1] largest,
 * [2] smallest pivot, [3] condition estimate trace(G) * ||G^-1|| (two inverse iterations on the factor; between
 * cond(G) and order * cond(G)).  1 <= order <= 1024 < len. */
#define IRA_AR_INFO_DOUBLES 4
#define IRA_AR_DENSE_GRAM 1   /* flags: form G = A^T A as a dense contraction on the FP64 matrix cores (v_mfma_f64_16x16x4_f64)
                               * instead of the O(order * len) lag-sum form: the cross-check of the default path.  gram, solve
                               * and refine of one fit take the same flags (layout of partial_dev). */
#define IRA_AR_WORKGROUP_SOLVE 2   /* flags (ira_ar_solve / ira_ar_refine): the 256-thread solve kernel also for order <= 64, where one wave
                                    * per element is the default since round 4 (A/B and cross-check: both give the same bits) */
int64_t ira_ar_partial_doubles(int32_t order, int32_t max_len);
/* The two halves of ira_ar_fit, callable separately: the MFMA Gram contraction, and reduce + Cholesky solve. */
int32_t ira_ar_gram(const float* x_dev, const double* x64_dev, const int64_t* xoff_dev,
                    const int32_t* len_dev, const double* divisor_dev, int32_t nb, int32_t max_len,
                    int32_t order, double* partial_dev, int32_t flags, void* stream);
int32_t ira_ar_solve(const double* partial_dev, const int32_t* len_dev, int32_t nb, int32_t max_len,
                     int32_t order, double ridge, double* gscratch_dev, double* coeffs_dev,
                     double* info_dev, int32_t flags, void* stream);
/* Rank-deficient fits.  For the elements ira_ar_solve flagged with status 1 (a Cholesky pivot was not positive: constant
 * segments, a few taps followed by digital silence, segments shorter than ~2 order) -- every other element is left alone --
 * G is eigen-decomposed (cyclic Jacobi) and the MINIMUM-NORM solution a = -V diag(1/lambda_k | 0) V^T r is written, dropping
 * directions with lambda_k <= rel_cut * lambda_max: what numpy.linalg.lstsq(A, y, rcond=None) returns for a rank-deficient A
 * (reference analyse/zplane.py:117).  info becomes [4, lambda_max, smallest kept lambda, rank]; a Gram matrix holding NaN or
 * infinity (the reference's lstsq raises LinAlgError there) gives status 3 and NaN coefficients.  scratch2_dev: nb * 2 *
 * order^2 doubles; order <= 512; rel_cut in (0, 1), 1e-12 sits just above the rounding noise of a float64 Gram matrix.
 * Runs after ira_ar_solve (ridge = 0, default lag-sum record: not with IRA_AR_DENSE_GRAM) on the same partial / coeffs /
 * info buffers, before ira_ar_refine. */
int32_t ira_ar_minnorm(const double* partial_dev, const int32_t* len_dev, int32_t nb, int32_t max_len, int32_t order,
                       double* scratch2_dev, double* coeffs_dev, double* info_dev, double rel_cut, void* stream);
/* Ill-conditioned fits beyond the reach of refinement: elements whose float64 Cholesky broke down (status 1) or whose
 * condition estimate info[3] exceeds cond_threshold (1e13: cond(G) eps is no longer small) are solved again with the normal
 * equations carried in DOUBLE-DOUBLE arithmetic: the p+1 lag sums accumulated from exact products of the samples (two_prod +
 * compensated sums, ~106 bits), G assembled from them, Cholesky and both triangular solves in the same arithmetic.  The
 * result is good to cond(G) x 1e-32 -- better than the reference's SVD-based lstsq (zplane.py:117, ~cond(A) eps), with
 * which it agrees to 2e-9 on tests/golden/ar_illcond.npz (500 Hz low-passed float32 response, cond(G) ~ 1e17, where the
 * float64 normal equations are off by 35 %).  info[0] becomes 5; a Gram matrix singular to 1e-26 of its trace keeps status
 * 1 for ira_ar_minnorm.  Runs after ira_ar_solve, before ira_ar_minnorm / ira_ar_refine, on the same partial / coeffs / info
 * buffers (lag-sum record only: not with IRA_AR_DENSE_GRAM).  ddpartial_dev: nb * ira_ar_exact_doubles(order, max_len, 0)
 * doubles, ddscratch_dev: nb * ira_ar_exact_doubles(order, max_len, 1).  Unflagged elements cost nothing but the launches. */
int64_t ira_ar_exact_doubles(int32_t order, int32_t max_len, int32_t which);
int32_t ira_ar_exact(const float* x_dev, const double* x64_dev, const int64_t* xoff_dev, const int32_t* len_dev,
                     const double* divisor_dev, int32_t nb, int32_t max_len, int32_t order, double ridge,
                     const double* partial_dev, double* ddpartial_dev, double* ddscratch_dev, double* coeffs_dev,
                     double* info_dev, double cond_threshold, void* stream);
/* Iterative refinement for ill-conditioned fits (ridge = 0 only).  The reference's SVD-based lstsq is accurate to about
 * cond(A) eps, the normal equations only to cond(A)^2 eps = cond(G) eps; elements whose condition estimate
 * info[3] > cond_threshold get `steps` (1..4) rounds of  a += G^-1 A^T (y - A a)  with the residual and
 * its gradient formed from the samples in float64 (corrected semi-normal equations): error ~ cond(A) eps again as long
 * as cond(A)^2 eps < 1.  Runs after ira_ar_solve with the same partial/coeffs/info buffers; grad_dev: nb *
 * ceil((max_len - order) / 4096) * (order + 1) doubles of scratch.  Unflagged elements cost nothing but the launch.
 * info[0] becomes 2 for refined elements. */
int32_t ira_ar_refine(const float* x_dev, const double* x64_dev, const int64_t* xoff_dev, const int32_t* len_dev,
                      const double* divisor_dev, int32_t nb, int32_t max_len, int32_t order, const double* partial_dev,
                      double* gscratch_dev, double* coeffs_dev, double* info_dev, double* grad_dev,
                      double cond_threshold, int32_t steps, int32_t flags, void* stream);
int32_t ira_ar_fit(const float* x_dev, const double* x64_dev, const int64_t* xoff_dev,
                   const int32_t* len_dev, const double* divisor_dev, int32_t nb, int32_t max_len, int32_t order, double ridge,
                   double* partial_dev, double* gscratch_dev, double* coeffs_dev, double* info_dev,
                   int32_t flags, void* stream);

/* All complex roots of npoly real polynomials given in DESCENDING powers, ncoef coefficients each
 * (Aberth-Ehrlich, float64).  Trailing coefficients with |c| < trail_eps are dropped first (reference
 * analyse/zplane.py:153-155), then numpy.roots conventions apply (leading zeros stripped, exact trailing
 * zeros are roots at 0).  roots_dev: npoly * (ncoef-1) interleaved (re, im); nroots_dev[e] = count found.
 * Replaces _roots_from_poly_descending, reference analyse/zplane.py:145-158.  Root order is unspecified. */
int32_t ira_poly_roots(const double* coeffs_dev, int32_t npoly, int32_t ncoef, double trail_eps,
                       double* roots_dev, int32_t* nroots_dev, void* stream);

/* b[e][n] = sum_k a[e][k] * s[n-k], n = 0..zero_order.  Replaces _derive_fir_numerator_from_ar, reference
 * analyse/zplane.py:123-142. */
int32_t ira_fir_numerator(const double* coeffs_dev, int32_t order, const float* x_dev,
                          const int64_t* xoff_dev, const int32_t* len_dev, const double* divisor_dev,
                          int32_t nb, int32_t zero_order, double* b_dev, void* stream);

/* ---- Direct transforms of smooth lengths -------------------------------------------------------------------------------
 * For n = 2^a 3^b 5^c that splits as n1*n2 with both factors <= 1024 (480000 = 640*750, 2^19 = 512*1024, ...) a
 * two-pass mixed-radix four-step transform replaces the three-pass Bluestein of ira_rfft_any / ira_band_irfft; every
 * job of a call has the same length n.  ira_fft_smooth_split: IRA_OK and the split, or IRA_E_UNSUPPORTED (then use
 * the Bluestein entry points; also when IRA_NO_SMOOTH_FFT is set in the environment).  Tables for the split:
 *   t1_dev[k] = exp(-2 pi i k/n1), k < n1;  t2_dev[k] = exp(-2 pi i k/n2), k < n2;  tf_dev[k] = exp(-2 pi i k/n), k < n2.
 * work_dev: nb * n complex f64.  All other arguments mean what they mean in ira_rfft_any / ira_band_irfft (same
 * reference call sites: frequency_response.py:204-213, filterplot.py:145-152, rt60bands.py:170-175,
 * group_delay.py:95-109).
 * Half-size modes (round 3), so that a channel's transforms never share a complex transform with ANOTHER channel's signal
 * and its results cannot depend on its neighbours in a batch (byte-identical records for any sharding):
 *   ira_rfft_smooth(..., interleave = 1): every job is ONE real signal of even length 2 n carried as the n-point complex
 *     sequence x[2m] + i x[2m+1] (x2off_dev[e] = xoff_dev[e] + 1; data_len / win_len, when given, count REAL samples; the
 *     second signal's length arrays must be NULL); spec_off_dev[e] receives its n + 1 bins.  zpair: n complex per job --
 *     or NULL (round 4, n1 even): the untangling is then part of the second pass (the intermediate is laid out in tiles
 *     of a row and its mirror row, so Z[k] and Z[n-k] meet in one workgroup) and no scratch or split pass exists.
 *   ira_band_irfft_smooth(..., half_out = 1): every job is ONE band (band record 2e; record 2e+1 is ignored) of a real
 *     signal of even length 2 n, computed by an n-point transform: the job's half spectrum has n + 1 bins, y1_off_dev[e]
 *     receives 2 n samples, y2_off_dev is ignored, spec_off2_dev must be NULL.
 *   ira_band_irfft_smooth(..., job_info_dev): nb * 4 int32 of device scratch, or NULL.  With it (round 4) NARROW jobs -- the
 *     hull of the job's band supports spans at most 9 n2 bins, e.g. the third-octave bands below ~800 Hz of a 10 s file --
 *     skip the first pass and the n-point work array altogether: their few non-zero bins are compacted once per job and
 *     the second pass sums the <= 2 x 9 terms of the pruned first pass per point in its input stage.  Which jobs are narrow
 *     is decided on the device from the mask records (job_info_dev[4e .. 4e+3] = {narrow, first bin, bins, terms}, an
 *     output for the curious).  Same results up to float64 rounding of a different summation order; NULL = every job
 *     takes both passes.
 *   ira_band_irfft_smooth(..., tile_part_dev) (round 5; may be NULL): the second pass also leaves the ENERGIES of the
 *     4096-sample tiles (counted from the end of each band signal) of what it writes, as partial sums per workgroup:
 *     job e, signal s (0: y1, 1: y2; a half_out job has signal 0 only), tile j, workgroup w at
 *     tile_part_dev[((e * 2 + s) * workgroups + w) * tiles + j], with (tiles, workgroups) = ira_band_tile_layout(n, half_out):
 *     nb * 2 * tiles * workgroups doubles.  ira_edc_fits(tile_part_dev ...) turns them into its tile totals and no longer
 *     reads every band signal twice (reference rt60bands.py:170-175 -> :272-321). */
int32_t ira_fft_smooth_split(int32_t n, int32_t* n1, int32_t* n2);
int32_t ira_band_tile_layout(int32_t n, int32_t half_out, int32_t* tiles, int32_t* workgroups);
int32_t ira_rfft_smooth(const float* x_dev, const int64_t* xoff_dev, int32_t n, int32_t nb, int32_t use_hann,
                        const void* t1_dev, const void* t2_dev, const void* tf_dev, double* work_dev,
                        double* spec_out_dev, const int64_t* spec_off_dev, const int64_t* x2off_dev,
                        const int64_t* spec_off2_dev, double* zpair_dev, const int64_t* zpair_off_dev,
                        const int32_t* data_len_dev, const int32_t* win_len_dev, const int32_t* data_len2_dev,
                        const int32_t* win_len2_dev, int32_t interleave, void* stream);
int32_t ira_band_irfft_smooth(const double* spec_dev, const int64_t* spec_off_dev, int32_t n, int32_t nb,
                              const double* band_params_dev, const double* freq_val_dev, const void* t1_dev,
                              const void* t2_dev, const void* tf_dev, double* work_dev, float* y_dev,
                              const int64_t* y1_off_dev, const int64_t* y2_off_dev, const int64_t* spec_off2_dev,
                              int32_t half_out, int32_t* job_info_dev, double* tile_part_dev, void* stream);

/* k-th smallest values (0-based ranks, clipped to the segment) of float64 segments values_dev + off_dev[e], count_dev[e]
 * long: out_dev[e*nranks + j] = sorted(segment)[ranks_dev[e*nranks + j]], 1 <= nranks <= 8 (radix select, exact).
 * The building block of numpy.median / numpy.percentile in summarise_group_delay_results_text, reference
 * analyse/group_delay.py:209-220 (the interpolation between neighbouring ranks stays on the host). */
int32_t ira_order_stats(const double* values_dev, const int64_t* off_dev, const int32_t* count_dev, int32_t nseg,
                        const int64_t* ranks_dev, int32_t nranks, double* out_dev, void* stream);

/* ---- Section 8f: diffusion / decorrelation per time window ----------------------------------------------------------
 * Element e: float32 signal at x_dev + xoff_dev[e] (already trimmed), nframes_dev[e] windows of `win` samples every
 * `hop` (16 <= win <= 8192, 1 <= max_lag <= 4096).  Per window, with w0 = w - mean(w) in float32 exactly as numpy
 * computes it (pairwise float32 sum):
 *   ac[e][f] = max_{lag=1..min(max_lag, win-2)} |sum_k w0[k] w0[k+lag]| / sum w0^2     (NaN if the energy <= 1e-20)
 *   ed[e][f] = fraction(|w0| > float32(thr_rms * rms)) [/ gauss_expected if gauss_expected >= 0], rms = float32
 *              sqrt(mean(w0*w0));  NaN if rms <= 1e-20 or 0 <= gauss_expected <= 1e-12.
 * Outputs float32 at out_off_dev[e] + f.  Replaces _windowed_max_abs_autocorr / _windowed_echo_density and the frame
 * loop of analyse_diffusion_for_channel, reference analyse/diffusion.py:139-159, :205-226, :264-276. */
int32_t ira_diffusion(const float* x_dev, const int64_t* xoff_dev, const int32_t* nframes_dev, int32_t nb,
                      int32_t max_frames, int32_t win, int32_t hop, int32_t max_lag, double thr_rms,
                      double gauss_expected, float* ac_dev, float* ed_dev, const int64_t* out_off_dev,
                      void* stream);

/* Stereo pair per element (left at loff_dev[e], right at roff_dev[e], same geometry): zero-lag Pearson correlation
 * and max |normalised cross-correlation| over lags -max_lag..+max_lag per window.  Replaces _windowed_corr0 /
 * _windowed_iacc_max and the frame loop of reference analyse/diffusion.py:162-202, :346-358. */
int32_t ira_diffusion_stereo(const float* x_dev, const int64_t* loff_dev, const int64_t* roff_dev,
                             const int32_t* nframes_dev, int32_t nb, int32_t max_frames, int32_t win, int32_t hop,
                             int32_t max_lag, float* corr0_dev, float* iacc_dev, const int64_t* out_off_dev,
                             void* stream);

/* ---- Section 8f, rank 2: native tap ingest (RIFF/WAVE 16-bit PCM as written by the reference's C++ recorder,
 * include/analysis/recorder.hpp:55-90) ----------------------------------------------------------------------------
 * ira_wav_probe: HOST call; walks the RIFF chunks of `path` and reports rate / channels / frames and the byte offset
 *   of the sample data.  IRA_OK for mono/stereo PCM16, IRA_E_UNSUPPORTED for any other valid WAV encoding (use the
 *   Python reader), IRA_E_IO if the file cannot be opened or is shorter than
 *   its header says, IRA_E_FORMAT if it is not RIFF/WAVE (scipy.io.wavfile.read raises FileNotFoundError / ValueError there).
 * ira_wav_read_pcm16: HOST call; reads frames*channels interleaved int16 into dst_host (ideally pinned memory).
 * ira_pcm16_to_channels: DEVICE; interleaved int16 -> planar float32 channels (out[c*frames + i]) with the
 *   reference's conversion x/32768 clipped to [-1, 1] (analyse/io.py:46-64, :98-113), or with mono_downmix (stereo
 *   only) the single channel 0.5*(L+R) in float32 (analyse/io.py:85-91).  Replaces scipy.io.wavfile.read +
 *   convert_wav_samples_to_float32 + get_analysis_channels for tap files; the upload carries 2 bytes per sample. */
int32_t ira_wav_probe(const char* path, int32_t* sample_rate, int32_t* channels, int64_t* frames,
                      int64_t* data_offset);
int32_t ira_wav_read_pcm16(const char* path, int64_t data_offset, int64_t frames, int32_t channels,
                           int16_t* dst_host);

/* The same for a whole group of files in ONE call, on `threads` host threads created and joined inside the call (a caller
 * with an interpreter lock releases it once per group instead of once per file).  status[i] = what ira_wav_probe /
 * ira_wav_read_pcm16 returns for file i; the return value is IRA_OK unless an argument is NULL / n < 0.  File i's payload
 * goes to dst_host + dst_off[i] (int16 elements).  Replaces the per-file loop around scipy.io.wavfile.read of reference
 * analyse/bundle.py:56-67 -> report.py -> io.py:200. */
int32_t ira_wav_probe_batch(const char* const* paths, int32_t n, int32_t threads, int32_t* status, int32_t* sample_rate,
                            int32_t* channels, int64_t* frames, int64_t* data_offset);
int32_t ira_wav_read_pcm16_batch(const char* const* paths, const int64_t* data_offset, const int64_t* frames,
                                 const int32_t* channels, const int64_t* dst_off, int16_t* dst_host, int32_t n,
                                 int32_t threads, int32_t* status);
int32_t ira_pcm16_to_channels(const int16_t* pcm_dev, int64_t frames, int32_t channels, int32_t mono_downmix,
                              float* out_dev, void* stream);
/* The same conversion for a GROUP of tap files in one launch (a bundle step holds 32 taps; one launch per tap was 27 % of a
 * step's device time).  File f: interleaved int16 at pcm_dev + src_off[f] (even element offset: 4-byte frame loads),
 * frames[f] frames of channels[f] in {1, 2} channels, mode[f] 1 = stereo mixed down to 0.5 * (L + R) in float32 (reference
 * analyse/io.py:85-91), 0 = planar channels; output at out_dev + dst_off[f] (channel c at + c * frames[f]).  max_frames >=
 * every frames[f] sizes the grid.  Replaces the per-file loop over scipy.io.wavfile.read + convert_wav_samples_to_float32
 * + get_analysis_channels of reference analyse/io.py:46-113, :181-221 for a whole group. */
int32_t ira_pcm16_to_channels_jobs(const int16_t* pcm_dev, const int64_t* src_off_dev, const int64_t* frames_dev,
                                   const int32_t* channels_dev, const int32_t* mode_dev, const int64_t* dst_off_dev,
                                   int32_t nfiles, int64_t max_frames, float* out_dev, void* stream);

/* ---- a8 alone: mask_dev[k] = the float32 mask value the band inverses multiply bin k with, k < nbins, for ONE band record
 * (band_params8: HOST array of 8 doubles, as in ira_band_irfft) on the axis float32(k * freq_val).  The reference's
 * _make_lowpass_mask / _make_highpass_mask / _make_bandpass_mask (analyse/rt60bands.py:116-167) evaluated by the very
 * device function the transforms inline -- for tests and for callers that want the mask itself. */
int32_t ira_band_mask_values(const double* band_params8, double freq_val, int64_t nbins, float* mask_dev, void* stream);

/* ira_host_pull: DEVICE kernel that reads PINNED (mapped) host memory over the PCIe link and writes HBM -- the batch upload
 *   without the copy engine, and for PCM16 with the conversion of io.py:46-64 (x/32768 clipped) in the same pass (no int16
 *   staging buffer in HBM).  host_src: host pointer of a pinned allocation (hipHostMalloc / torch pin_memory), 16-byte
 *   aligned; count samples; format 0 = float32 copied as is, 1 = mono int16 -> float32; out_dev 16-byte aligned;
 *   workgroups = grid size (0 = 8: just enough reads in flight to fill a Gen5 x16 link; more only crowd the fabric queues
 *   the analysis kernels' HBM reads go through).  Throughput equals hipMemcpyAsync's within noise: an
 *   option, not the default.  IRA_E_UNSUPPORTED if host_src is not mapped host memory. */
int32_t ira_host_pull(const void* host_src, int64_t count, int32_t format, float* out_dev, int32_t workgroups,
                      void* stream);

/* ---- Section 8f, rank 4: sweep deconvolution (reference analyse/deconvolve.py:124-193) -----------------------------------
 * The transforms are ira_rfft_any / ira_rfft_smooth (zero-padded to n_fft = next power of two, no window) and
 * ira_band_irfft / ira_band_irfft_smooth (all-pass mask).  In between and after:
 * ira_deconv_divide: element e: Y (half spectrum of nfft[e]/2+1 bins at yspec_dev + 2*yspec_off[e], overwritten) becomes
 *   H = Y conj(X) / (|X|^2 + eps) with X at xspec_dev + 2*xspec_off[e] (elements may share one sweep spectrum) and
 *   eps = regularization_relative * max(1e-30, max_k |X_k|^2), |X|^2 formed as hypot(re, im)^2 like numpy.abs(X)**2
 *   (deconvolve.py:150-166).  pmax_dev: nb doubles of scratch (receives max |X|^2 per element).
 * ira_deconv_finish: element e is one channel of length n_out[e] at h_dev + h_off[e], group[e] < ngroups its FILE:
 *   remove_dc: h -= float32(mean(h)) per channel; normalise_peak: every channel of a file is multiplied by
 *   float32(target_peak / max |h| over the file) unless that peak is 0 (deconvolve.py:176-189, :100-106).  mean_dev: nb
 *   floats, peak_bits_dev: ngroups uint32 of scratch. */
int32_t ira_deconv_divide(double* yspec_dev, const int64_t* yspec_off_dev, const double* xspec_dev,
                          const int64_t* xspec_off_dev, const int32_t* nfft_dev, int32_t nb, int32_t max_nfft,
                          double regularization_relative, double* pmax_dev, void* stream);
int32_t ira_deconv_finish(float* h_dev, const int64_t* h_off_dev, const int32_t* n_out_dev, const int32_t* group_dev,
                          int32_t nb, int32_t ngroups, int32_t max_len, int32_t remove_dc, int32_t normalise_peak,
                          double target_peak, float* mean_dev, uint32_t* peak_bits_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IRA_H_ */
