#!/bin/bash
# tools/ab.sh -- ONE parametrised A-B-A driver for kernel experiments on one GPU box (replaces the 29 one-off tools/r03_ab*.sh
# of round 3; their logs are kept under profiles/r03_*_ab.txt).
#
#   tools/ab.sh NAME [-t "pytest -k expression"] [-T "test files"] [-r passes] [-e "ENV=1 ENV2=x"] -b "bench command" VARIANT...
#
#   NAME      log goes to gpurun_out/ab_NAME.log
#   -t / -T   parity gate first: python3 -m pytest <files> -m gpu -q -x -k <expr>; a red gate stops the run (only a parity-green
#             library may be timed)
#   -b        the timed command (e.g. "python3 tools/bench_attn.py"), run once per variant and pass
#   -e        environment for the timed command (HEADLINE=1 ONLY64=1 REPS=5 ...)
#   VARIANT   "" / "tree" = the in-tree librsvld_hip.so, anything else = tools/ablate/librsvld_<VARIANT>.so
#             (built by tools/build_variant.sh <VARIANT> <file.hip> <hipcc flags>), selected through RSVLD_LIB
# Passes alternate the variants (A B A B ...) so that clock drift of the box shows up as a difference between passes.
set -u
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
name=$1; shift
kexpr=""; tfiles="tests/test_gpu_kernels.py"; passes=2; envs=""; bench=""
while getopts "t:T:r:e:b:" o; do
  case $o in t) kexpr=$OPTARG;; T) tfiles=$OPTARG;; r) passes=$OPTARG;; e) envs=$OPTARG;; b) bench=$OPTARG;; *) exit 2;; esac
done
shift $((OPTIND - 1))
[ -n "$bench" ] || { echo "ab.sh: -b \"bench command\" is required"; exit 2; }
[ $# -gt 0 ] || set -- tree
mkdir -p gpurun_out; log=gpurun_out/ab_$name.log; : > "$log"
if [ -n "$kexpr" ]; then
  timeout -k 10 600 python3 -m pytest $tfiles -m gpu -q -x -k "$kexpr" >> "$log" 2>&1 || { echo "TESTS FAILED" >> "$log"; tail -30 "$log"; exit 1; }
fi
for rep in $(seq 1 "$passes"); do
  for v in "$@"; do
    lib=""; [ "$v" != "tree" ] && [ -n "$v" ] && lib=$R/tools/ablate/librsvld_$v.so
    echo "== variant: ${v:-tree} (pass $rep)" >> "$log"
    env $envs ${lib:+RSVLD_LIB=$lib} timeout -k 10 300 $bench >> "$log" 2>&1 || { echo "BENCH FAILED (variant $v)" >> "$log"; tail -20 "$log"; exit 1; }
  done
done
tail -60 "$log"
