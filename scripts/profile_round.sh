#!/bin/bash
# The profiling recipe of profiles/README.md for one round: bash scripts/profile_round.sh r3   (on the GPU box)
# Raw output lands in gpurun_out/<tag>_*; `python scripts/summarize_profile.py <tag>` then condenses it into profiles/.
set -e
tag=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "$R"
mkdir -p "$O"
timeout -k 10 420 python bench.py > "$O/${tag}_bench.json" 2> "$O/${tag}_bench.err"
echo "[profile] plain bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_trace" -o "$tag" -- python bench.py --no-cpu-baseline --no-decode > "$O/${tag}_bench_under_rocprof.json" 2> "$O/${tag}_trace.err"
echo "[profile] kernel trace done"
C="python bench.py --no-cpu-baseline --no-decode --steps 4 --warmup 2"
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/${tag}_pmc_fetch" -o "$tag" -- $C --sweep-ms 1,128 > /dev/null 2> "$O/${tag}_pmc_fetch.err"
echo "[profile] FETCH_SIZE pass done"
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/${tag}_pmc_write" -o "$tag" -- $C --sweep-ms 1,128 > /dev/null 2> "$O/${tag}_pmc_write.err"
echo "[profile] WRITE_SIZE pass done"
timeout -s KILL 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d "$O/${tag}_pmc_mfma" -o "$tag" -- $C --sweep-ms 32,128,512,1024,4096 > /dev/null 2> "$O/${tag}_pmc_mfma.err"
echo "[profile] MFMA pass done"
