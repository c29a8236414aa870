#!/bin/bash
# One GPU session (gpurun): ./tools/gpu_session.sh <name> <step> [<step> ...]; outputs under gpurun_out/<name>/.
# Steps: mall | tests:<pytest args> | bench:<workload>[:<extra bench args>] | sharded:<workload>[:mode] | full | pmc:<workload> | stats:<workload>
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
name="$1"; shift
out="gpurun_out/$name"; mkdir -p "$out"
export TMPDIR=/tmp
for step in "$@"; do
  kind="${step%%:*}"; rest="${step#*:}"
  echo "=== $step ($(date +%T))"
  case "$kind" in
    mall) ./tools/micro_mall.bin > "$out/micro_mall.txt" 2>&1; tail -12 "$out/micro_mall.txt" ;;
    tests) timeout 1500 python -m pytest $rest -x -q -p no:cacheprovider --timeout 300 > "$out/pytest_$(echo "$rest" | tr -c 'a-zA-Z0-9' '_' | cut -c1-40).log" 2>&1; tail -5 "$out"/pytest_*.log ;;
    bench) wl="${rest%%:*}"; extra=""; [ "$rest" != "$wl" ] && extra="${rest#*:}"
           timeout 600 python bench.py --workload "$wl" --no-extras --no-cpu --steps 5 --warmup 2 $extra > "$out/bench_$wl.json" 2> "$out/bench_$wl.err"; echo "rc=$?"; cut -c1-1500 "$out/bench_$wl.json"; tail -3 "$out/bench_$wl.err" ;;
    sharded) wl="${rest%%:*}"; mode="shuffle"; [ "$rest" != "$wl" ] && mode="${rest#*:}"
           PLX_BENCH_FORCE_SHARDED=1 PLX_Q3_MODE="$mode" timeout 600 python bench.py --workload "$wl" --no-extras --steps 3 --warmup 1 > "$out/sharded_${wl}_$mode.json" 2> "$out/sharded_${wl}_$mode.err"; echo "rc=$?"; cut -c1-1500 "$out/sharded_${wl}_$mode.json"; tail -3 "$out/sharded_${wl}_$mode.err" ;;
    full) timeout 900 python bench.py --steps 20 --warmup 5 > "$out/bench_full.json" 2> "$out/bench_full.err"; echo "rc=$?"; cut -c1-600 "$out/bench_full.json"; tail -5 "$out/bench_full.err" ;;
    stats) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_$rest" -o "$rest" -- python "$OLDPWD/bench.py" --workload "$rest" --no-extras --no-cpu --steps 5 --warmup 2 > "$OLDPWD/$out/stats_$rest.json" 2> "$OLDPWD/$out/stats_$rest.err"); echo "rc=$?" ;;
    pmc) for ctr in FETCH_SIZE WRITE_SIZE; do (cd /tmp && PLX_BENCH_VERIFY=0 timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d "$OLDPWD/$out/pmc_${rest}_$ctr" -o "$rest" -- python "$OLDPWD/bench.py" --workload "$rest" --no-extras --no-cpu --steps 3 --warmup 2 > /dev/null 2> "$OLDPWD/$out/pmc_${rest}_$ctr.err"); echo "rc=$?"; done ;;
    *) echo "unknown step $step" ;;
  esac
done
echo "=== done ($(date +%T))"
