#!/bin/bash
# Config 3 (third-octave RT60 bank + waterfall, 256 x 10 s per step) and the report step with and without the band inverses'
# tile energies (IRA_BAND_TILE_ENERGIES), alternating on one box.   bash tools/experiments/r5_tile_energy_ab.sh
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in 3 report; do
    for on in 1 0; do
      IRA_BAND_TILE_ENERGIES=$on timeout -k 10 400 python3 $R/bench.py --config $cfg --no-cpu-baseline --variants resident --literal-steps 0 --roofline-steps 2 > /tmp/te.json 2> /tmp/te.err || echo failed
      python3 -c "
import json; d=json.load(open('/tmp/te.json')); c=d['device_ms_per_step_by_call']
print('config $cfg tile energies $on rep $rep: value', round(d['value']), 'resident', round(d['value_resident']), 'ms/step', round(d['ms_per_step'],2), '| edc_fits', round(c.get('ira_edc_fits',0),3), 'band_irfft_smooth', round(c.get('ira_band_irfft_smooth',0),3), 'device total', round(d['device_ms_per_step'],2))"
    done
  done
done
