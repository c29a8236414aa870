#!/bin/bash
# measurement session for the wide-key aggregation pass: ablations + SQ counters (gpurun: ./tools/exp_wide.sh <name>)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out="gpurun_out/$1"; mkdir -p "$out"; export TMPDIR=/tmp
run() { tag="$1"; shift; env "$@" PLX_BENCH_VERIFY=0 timeout 300 python bench.py --workload cfg3w --no-extras --no-cpu --steps 4 --warmup 2 > "$out/$tag.json" 2> "$out/$tag.err"; echo "$tag rc=$?";
        python - "$out/$tag.json" <<'P'
import json,sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("[bench] full record: ")]
d=json.loads(l[-1][21:]) if l else {}
print("   ms", d.get("ms_per_step"), {k:v["avg_us"] for k,v in (d.get("kernels") or {}).items() if v["avg_us"]>100})
P
}
run base X=1
run abl8 PLX_PART_ABLATE=8
run abl4 PLX_PART_ABLATE=4
run abl12 PLX_PART_ABLATE=12
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_WAVES"; do
  tag=$(echo "$pass" | cut -d' ' -f1)
  (cd /tmp && PLX_BENCH_VERIFY=0 timeout 400 rocprofv3 --pmc $pass --kernel-trace -d "$OLDPWD/$out/pmc_$tag" -o w -- python "$OLDPWD/bench.py" --workload cfg3w --no-extras --no-cpu --steps 2 --warmup 1 > /dev/null 2> "$OLDPWD/$out/pmc_$tag.err"); echo "pmc $tag rc=$?"
  python - "$out/pmc_$tag" <<'P'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
if not f: print("no csv"); sys.exit()
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]
    if "part" in k: acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {c:(round(sorted(x)[len(x)//2]/1e6,2)) for c,x in v.items()}, "(median per launch, millions)")
P
done
