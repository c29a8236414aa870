#!/bin/bash
# ./tools/exp_quick.sh <name> <workload> [ENV=V ...]: one short bench run of a workload (no extras, no CPU leg, no host check), its step time and kernel table
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out="gpurun_out/$1"; mkdir -p "$out"; wl="$2"; shift 2
tag="${wl}_$(echo "$*" | tr -c 'a-zA-Z0-9=' '_' | cut -c1-40)"
env "$@" PLX_BENCH_VERIFY=${PLX_BENCH_VERIFY:-0} timeout 400 python bench.py --workload "$wl" --no-extras --no-cpu --steps ${STEPS:-4} --warmup 2 > "$out/$tag.json" 2> "$out/$tag.err"; echo "$tag rc=$?"
python - "$out/$tag.json" <<'P'
import json,sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("[bench] full record: ")]
d=json.loads(l[-1][21:]) if l else {}
print("   ms", d.get("ms_per_step"), "one_shot", d.get("one_shot_ms"), "cold", d.get("cold_first_step_ms"), "verified", (d.get("verified") or {}).get("ok"), {k:v["avg_us"] for k,v in (d.get("kernels") or {}).items() if v["avg_us"]>80})
P
tail -2 "$out/$tag.err"
