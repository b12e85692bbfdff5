#!/usr/bin/env python3
"""profiles/*_pmc_traffic.json from two rocprofv3 --pmc passes over tools/kernels_once.py
(FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, TCC counter budget).

    python3 tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Units and the gfx950 correction follow the guide's HBM section: both counters are in KiB;
FETCH_SIZE tallies the 128 B requests of wide streaming reads at 64 B, so it is doubled."""
import csv
import json
import sys

KEYS = [('k_gae_pipe', 'k_gae_pipe T=2048 N=4096'), ('k_gather_wide', 'k_gather_wide atari B=1024'),
        ('k_gather_narrow', None)]


def per_kernel(path, counter):
  out = {}
  for r in csv.DictReader(open(path)):
    if r['Counter_Name'] != counter:
      continue
    out.setdefault((r['Kernel_Name'], int(r['Grid_Size'])), []).append(float(r['Counter_Value']) * 1024.0)
  return out


def main():
  fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
  res = {}
  for (name, grid), vals in sorted(fetch.items()):
    for sub, key in KEYS:
      if sub not in name:
        continue
      if key is None:     # the narrow gather runs at two sizes: the Atari records' 12 B tail and Pendulum rows
        key = 'k_gather_narrow pendulum B=65536' if grid >= 65536 * 4 else 'k_gather_narrow atari tail B=1024'
      f = sum(vals) / len(vals)
      w = write.get((name, grid), [0.0])
      w = sum(w) / len(w)
      res[key] = {'FETCH_SIZE_bytes_raw': f, 'fetch_bytes_corrected_x2': 2 * f, 'WRITE_SIZE_bytes': w,
                  'hbm_bytes': 2 * f + w, 'launches': len(vals), 'kernel': name[:120]}
  json.dump(res, open(sys.argv[3], 'w'), indent=1)
  print(json.dumps({k: int(v['hbm_bytes']) for k, v in res.items()}, indent=1))


if __name__ == '__main__':
  main()
