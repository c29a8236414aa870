"""Build-time guard: the pre-instantiated (AOT) hot kernels must not spill to scratch memory.  (Passing kernel arguments
by reference into the inlined kernel body once turned them into stack copies: the partition scatter kernel spilled
932 B/lane and ran 8x slower.)  Compiles the two kernel files with -Rpass-analysis=kernel-resource-usage; no GPU needed."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "polars_amd", "csrc")


def resource_usage(src):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-munsafe-fp-atomics",
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.devnull]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC, timeout=900).stderr
    res, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = t.split(": ", 1)[1]
            res[cur] = {}
        elif cur and ": " in t:
            k, v = t.split(": ", 1)
            res[cur][k.strip()] = v.strip()
    return res


def test_aot_kernels_do_not_spill():
    checked = 0
    for src in ("kernels_fused.hip", "kernels_partition.hip"):
        for name, r in resource_usage(src).items():
            if "StatProg" not in name:
                continue          # the generic interpreter's kernels are not the hot path
            checked += 1
            # the third-generation scatter with 12-byte records and 8192-row tiles sits exactly at its 128-register budget: ONE loop-invariant
            # register lives in scratch (stored in the prologue, reloaded once per round at a point where no column load is outstanding --
            # checked in the ISA); anything beyond that is the regression this test exists for
            # (the instantiations with the per-row check of narrowed values compiled in -- last template flag true -- may spill a few registers more in the two
            # shapes that sit at the budget: they run only until a predicate-free scan has verified the assumed bounds, engine.cpp mark_sources_verified)
            checked_form = "part3_scatter_kernel" in name and re.search(r"ELb[01]ELb1EEEv", name) is not None
            allowed = 48 if checked_form else 8 if "part3_scatter_kernel" in name else 0
            assert int(r["ScratchSize [bytes/lane]"]) <= allowed, (name, r)
            assert int(r["VGPRs"]) <= 256, (name, r)
            if "part2_" in name or "part3_" in name:       # 1024-thread workgroups (16 waves per CU): 128 VGPRs at most
                assert int(r["VGPRs"]) <= 128, (name, r)
    assert checked >= 33, checked


def test_sort_join_datagen_kernels_do_not_spill():
    """The radix-sort, hash-join and generator kernels keep everything in registers / LDS (the sort scatter stages its tile in
    exactly 64 KB of LDS: two workgroups per CU)."""
    checked = 0
    for src in ("kernels_sort.hip", "kernels_join.hip", "kernels_datagen.hip"):
        for name, r in resource_usage(src).items():
            checked += 1
            assert int(r["ScratchSize [bytes/lane]"]) == 0, (name, r)
            if "radix_scatter" in name:
                assert int(r["LDS Size [bytes/block]"]) <= 80 * 1024, (name, r)
    assert checked >= 15, checked


def test_parquet_kernels_do_not_spill_and_fit_two_snappy_workgroups_per_cu():
    """The Parquet decode kernels keep their state in registers; the Snappy workgroup's LDS (window, node tables, pointers, elements)
    stays under 80 KB so that two streams share a CU's 160 KB."""
    res = resource_usage("kernels_parquet.hip")
    assert len(res) >= 8
    for name, r in res.items():
        assert int(r["ScratchSize [bytes/lane]"]) == 0, (name, r)
        if "pq_snappy" in name:
            assert int(r["LDS Size [bytes/block]"]) <= 80 * 1024, (name, r)
            assert int(r["VGPRs"]) <= 256, (name, r)          # two 256-thread workgroups per CU = two waves per SIMD: 256 registers each at most
    assert sum("pq_snappy" in n for n in res) == 3           # generation 2 (the default since round 3) with and without its phase clock, generation 1 (PLX_SNAPPY_KERNEL=1)


def test_string_group_by_kernels_do_not_spill():
    """kernels_strgroup.hip: 1024-thread workgroups (128 registers at most); the scatter's first version with batched LDS reads spilled 648 B / lane.
    (The <true> instantiation is the PLX_STRGROUP_TIMING build with its phase clocks: diagnostics, allowed a few spilled registers.)"""
    res = resource_usage("kernels_strgroup.hip")
    assert len(res) >= 6
    for name, r in res.items():
        timing_build = "strgroup_scatter_kernelILb1" in name
        assert int(r["ScratchSize [bytes/lane]"]) <= (64 if timing_build else 0), (name, r)
        assert int(r["VGPRs"]) <= 128, (name, r)


def test_no_kernel_uses_flat_memory_instructions(tmp_path):
    """Every memory access of every kernel in the built library is a global_*, ds_* or scalar instruction -- never flat_*.  A flat access is counted by vmcnt AND
    lgkmcnt, so the wait behind it also waits for every global load in flight; it appears when the compiler cannot tell the address space: a `volatile` access
    through a pointer derived from the dynamic LDS base (the LDS slot protocols used those: the wide-key aggregation's slot search was 15 of 18.5 ms), or a pointer
    made from a 64-bit integer that was loaded from a descriptor (the Parquet page descriptors / Snappy jobs: 126 flat instructions in pq_snappy).  dev.hpp lds_ld /
    lds_st and parquet_device.hpp PQ_GPTR are the two idioms that avoid them."""
    import shutil
    lib = os.path.join(ROOT, "polars_amd", "libpolars_amd.so")
    assert os.path.exists(lib), "build the library first (__graft_entry__.build)"
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(lib, work / "lib.so")
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=work, capture_output=True, text=True, timeout=300)      # writes lib.so.<n>.hipv4-amdgcn-amd-amdhsa--gfx950
    objs = sorted(f for f in os.listdir(work) if f.endswith("gfx950"))
    assert len(objs) >= 10, objs
    kernels, offenders = 0, {}
    for f in objs:
        name = None
        for line in subprocess.run([objdump, "-d", f], cwd=work, capture_output=True, text=True, timeout=600).stdout.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
            if m:
                name = m.group(1); kernels += 1
            elif name and re.search(r"\bflat_(load|store|atomic)", line):
                offenders[name] = offenders.get(name, 0) + 1
    assert kernels >= 200, kernels
    assert not offenders, offenders


def test_jit_wide_key_aggregation_has_no_scratch_and_no_flat_access(tmp_path, monkeypatch):
    """The run-time compiled kernels of the two-column-key group-by (bench.py cfg3w): the LDS aggregation pass keeps two chunks of 5- or 6-dword records in
    flight next to its slot search -- three spilled -- and reads its table through ds_* instructions only (plx_jit_selftest + PLX_JIT_DUMP_DIR: no GPU needed)."""
    import sys
    sys.path.insert(0, ROOT)
    import polars_amd as pl
    from test_jit_cpu import ph
    monkeypatch.setenv("PLX_JIT_DUMP_DIR", str(tmp_path))
    t = pl.DataFrame([ph("id", pl.Int64), ph("id2", pl.Int64), ph("v", pl.Float64)])
    t.lazy().group_by("id", "id2").agg(pl.col("v").sum().alias("s"), pl.len().alias("n")).jit_selftest()
    objs = [f for f in os.listdir(tmp_path) if f.startswith("part3_agg") and f.endswith(".hsaco")]
    assert objs, os.listdir(tmp_path)
    for f in objs:
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f], cwd=tmp_path, capture_output=True, text=True, timeout=120).stdout
        assert int(re.search(r"\.vgpr_spill_count:\s*(\d+)", notes).group(1)) == 0, notes
        assert int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", notes).group(1)) == 0, notes
        isa = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", f], cwd=tmp_path, capture_output=True, text=True, timeout=120).stdout
        assert "ds_cmpst_rtn_b32" in isa and "ds_read_b128" in isa and "scratch_" not in isa and not re.search(r"\bflat_(load|store|atomic)", isa)
