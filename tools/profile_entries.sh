#!/bin/bash
# Per-configuration rocprofv3 evidence for bench.py's roofline entries (run on the GPU box):
#   one --kernel-trace --stats run per entry (the process holds that entry's launches and nothing else), then two
#   --pmc passes (FETCH_SIZE, WRITE_SIZE: they cannot share a pass) for the HBM-bound entries.
# Output: gpurun_out/prof_r03/ (TRL_PROF_ROUND names the round)/<entry>_kernel_stats.csv, <entry>_fetch.csv, <entry>_write.csv; copy the kernel
# stats into profiles/ and run tools/pmc_traffic.py over the counter files.
#   bash tools/profile_entries.sh [entry ...]
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/prof_${TRL_PROF_ROUND:-r03}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ENTRIES=("$@")
if [ ${#ENTRIES[@]} -eq 0 ]; then
  ENTRIES=(gae_cold_blocked gae_warm_blocked gae_cold_tn gae_warm_tn gather_wide_1024 gather_wide_256 gather_narrow_65536 gather_narrow_128 learn_kernel learn_kernel_bf16x6 learn_epoch learn_epoch_256)
fi
for e in "${ENTRIES[@]}"; do
  d="$OUT/raw_$e"
  rm -rf "$d"
  timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d "$d" -o "$e" --output-format csv -- python3 "$ROOT/bench.py" --only "$e" > "$OUT/${e}_line.json" 2> "$OUT/${e}.err" || { echo "kernel trace of $e failed"; tail -3 "$OUT/${e}.err"; exit 1; }
  f=$(find "$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${e}_kernel_stats.csv"
  echo "== $e"; head -4 "$OUT/${e}_kernel_stats.csv" | cut -c1-200
  case "$e" in
    gae_cold_*|gather_*)
      for c in FETCH_SIZE WRITE_SIZE; do
        dd="$OUT/raw_${e}_$c"
        rm -rf "$dd"
        timeout -s KILL 170 rocprofv3 --pmc $c -d "$dd" -o "$e" --output-format csv -- python3 "$ROOT/bench.py" --only "$e" --kernel-iters 6 > /dev/null 2> "$OUT/${e}_$c.err" || { echo "pmc pass $c of $e failed"; tail -3 "$OUT/${e}_$c.err"; exit 1; }
        f=$(find "$dd" -name "*counter_collection.csv" | head -1)
        [ -n "$f" ] && cp "$f" "$OUT/${e}_$c.csv"
      done
      ;;
  esac
  rm -rf "$OUT"/raw_"$e"*
done
echo done
