#!/usr/bin/env python3
"""profiles/r0N_pmc_traffic.json from the rocprofv3 --pmc passes of tools/profile_entries.sh: per bench entry
(`bench.py --only <entry>`), the HBM bytes per launch of that entry's kernel.  FETCH_SIZE and WRITE_SIZE cannot
share a pass (MI355X_MICROARCH.md, TCC counter budget), hence two files per entry.

    python3 tools/pmc_traffic.py gpurun_out/prof_r02 profiles/r02_pmc_traffic.json

Units and the gfx950 correction follow the guide's HBM section: both counters are in KiB; FETCH_SIZE tallies the
128 B requests of wide streaming reads at 64 B, so it is doubled (uncalibrated for the narrow gather's 4 B-per-lane
record reads: there the raw and the doubled figure are both kept)."""
import csv
import glob
import json
import os
import sys

KERNEL_OF = {'gae': 'k_gae_pipe', 'gather_wide': 'k_gather_wide', 'gather_narrow': 'k_gather_narrow'}


def mean_counter(path, counter, kernel):
  vals = [float(r['Counter_Value']) * 1024.0 for r in csv.DictReader(open(path))
          if r['Counter_Name'] == counter and kernel in r['Kernel_Name']]
  return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
  src, out = sys.argv[1], sys.argv[2]
  res = {}
  for fpath in sorted(glob.glob(os.path.join(src, '*_FETCH_SIZE.csv'))):
    entry = os.path.basename(fpath)[:-len('_FETCH_SIZE.csv')]
    wpath = os.path.join(src, entry + '_WRITE_SIZE.csv')
    kernel = next(v for k, v in KERNEL_OF.items() if entry.startswith(k))
    f, nf = mean_counter(fpath, 'FETCH_SIZE', kernel)
    w, nw = mean_counter(wpath, 'WRITE_SIZE', kernel) if os.path.exists(wpath) else (None, 0)
    if f is None or w is None:
      continue
    # The x2 of the guide is for 128 B requests of wide streaming reads tallied at 64 B.  The narrow gather's reads are
    # single 64 B requests (one padded record each): its raw count is already bytes -- calibrated in round 3 on records
    # padded to 64 B, where 65,536 random rows fetch 4.48 MB raw = 64 B per row + the 0.5 MB of indices
    # (tools/narrow_pad_experiment.sh); doubled, that launch would have fetched twice what it can touch.
    narrow = entry.startswith('gather_narrow')
    res[entry] = {'FETCH_SIZE_bytes_raw': f, 'fetch_bytes_corrected_x2': 2 * f, 'WRITE_SIZE_bytes': w,
                  'hbm_bytes': (f if narrow else 2 * f) + w, 'fetch_correction': 'none (64 B requests)' if narrow else 'x2 (128 B requests tallied at 64 B)',
                  'launches': min(nf, nw), 'kernel': kernel,
                  'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over `bench.py --only %s --kernel-iters 6`' % entry}
  json.dump(res, open(out, 'w'), indent=1)
  print(json.dumps({k: int(v['hbm_bytes']) for k, v in res.items()}, indent=1))


if __name__ == '__main__':
  main()
