# Round 4h: the batch floor of `squishrs pack` (SQUISH_BATCH_FLOOR, bytes) on a compressible 2,000 MiB tree (tools/soak_cli.py) and on
# 8 GiB of random 256 KiB files (tools/bench_cli.py: the end_to_end.big shape).  One box, back to back.
set -e
for f in 1073741824 2147483648 4294967296; do
  echo "== floor $f, compressible 2000 MiB"; SQUISH_BATCH_FLOOR=$f timeout -k 10 400 python tools/soak_cli.py 2000 2>&1 | tail -2
done
for f in 1073741824 4294967296; do
  echo "== floor $f, random 32768 x 256 KiB"; SQUISH_BATCH_FLOOR=$f timeout -k 10 400 python tools/bench_cli.py 32768 262144 2>&1 | tail -3 | cut -c1-700
done
