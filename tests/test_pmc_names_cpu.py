"""The counter summaries under profiles/ are keyed by the name the library's HIP-event tracer gives a launch (bench.py pairs a counter figure with a timed
kernel only on an exact match of that name): tools/pmc_summarise.py derives the same name from the kernel SYMBOL rocprofv3 reports.  This pins the mapping
for the kernel families of the bench workloads, on symbols copied from profiles/*/*_kernel_stats.csv, and checks that in every committed summary of the current
round EVERY kernel that takes >= 3 % of a step's kernel time has a counter figure under the name the same session's bench line gave it."""
import csv
import glob
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pmc_summarise", os.path.join(ROOT, "tools", "pmc_summarise.py"))
pmc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pmc)

CASES = [
    ("void plx::k::fused_scan_kernel<plx::k::StatProg<3>, plx::k::LdsAggSink>(plx::fused::Shape, plx::fused::Args, plx::fused::LdsAggParams)", "fused_scan_ldsagg_static#3"),
    ("void plx::k::fused_scan_kernel<plx::k::StatProg<8>, plx::k::DirectProbeAggSink>(plx::fused::Shape, plx::fused::Args, plx::fused::DirectJoinTable)", "fused_scan_direct_probe_agg_static#8"),
    ("void plx::k::fused_scan_kernel<plx::k::StatProg<7>, plx::k::DirectBuildSink>(plx::fused::Shape, plx::fused::Args, plx::fused::DirectJoinTable)", "fused_scan_direct_build_static#7"),
    ("void plx::k::part3_scatter_kernel<plx::k::StatProg<4>, 1, 4, 2, false>(plx::fused::Shape, plx::fused::Args, plx::k::PartPlan2, plx::k::ScatterParams2)", "part3_scatter[#4,d,t4,p2]"),
    ("void plx::k::part3_scatter_kernel<plx::k::StatProg<4>, 0, 2, 1, false>(plx::fused::Shape, plx::fused::Args, plx::k::PartPlan2, plx::k::ScatterParams2)", "part3_scatter[#4,h,t2,p1]"),
    ("void plx::k::part3_scatter_kernel<plx::k::StatProg<4>, 0, 3, 1, true>(plx::fused::Shape, plx::fused::Args, plx::k::PartPlan2, plx::k::ScatterParams2)", "part3_scatter[#4,h,t3,p1,hot]"),
    ("void plx::k::part3_scatter_kernel<plx::k::StatProg<4>, 1, 3, 2, true>(plx::fused::Shape, plx::fused::Args, plx::k::PartPlan2, plx::k::ScatterParams2)", "part3_scatter[#4,d,t3,p2,hot]"),
    ("plx::k::probe_pass_kernel(unsigned int const*, unsigned int const*, unsigned long long const*, unsigned int const*, unsigned long long const*, unsigned long long, unsigned long long, unsigned int, unsigned int, unsigned int, plx::k::ProbeHashedBuild, unsigned int*, unsigned int*)", "probe_pass_lds"),
    ("void plx::k::fused_scan_kernel<plx::k::StatProg<7>, plx::k::JoinBuildSink>(plx::fused::Shape, plx::fused::Args, plx::fused::JoinAggTable)", "fused_scan_join_build_static#7"),
    ("void plx::k::part3_scatter_kernel<plx::k::StatProg<12>, 1, 4, 3, false>(plx::fused::Shape, plx::fused::Args, plx::k::PartPlan2, plx::k::ScatterParams2)", "probe_scatter[#12,d,t4,p3]"),
    ("void plx::k::part2_agg_kernel<plx::k::StatProg<5>, 1, 0>(plx::k::PartPlan2, plx::k::AggParams2)", "part_agg_lds[#5,d,p0]"),
    ("void plx::k::(anonymous namespace)::strgroup_scatter_kernel<false>(plx::k::(anonymous namespace)::SgScatter)", "strgroup_scatter"),
    ("plx::k::(anonymous namespace)::strgroup_agg_kernel(plx::k::(anonymous namespace)::SgAgg)", "strgroup_agg_lds"),
    ("plx::k::(anonymous namespace)::sg_chunk_place_kernel(unsigned int const*, long, unsigned int, unsigned long long const*, unsigned int*, unsigned int*)", "part2_chunk_sort"),
    ("plx::k::direct_pairs_compact_kernel(plx::fused::DirectJoinTable, long, int, int, unsigned long long*, unsigned long long*, unsigned int*, unsigned long long*)", "table_compact"),
    ("void plx::k::datagen_uniform_kernel<long>(long, unsigned long, unsigned int, long, long, double, long*)", "datagen_uniform_i64"),
    ("void plx::k::part3_scatter_kernel<plx::k::StatProg<13>, 0, 1, 1, false>(plx::fused::Shape, plx::fused::Args, plx::k::PartPlan2, plx::k::ScatterParams2)", "part3_scatter[#13,h,t1,p1]"),
    # (round 5: a sixth template flag -- the per-row check of narrowed values compiled in or not; both forms are one tracer name)
    ("void plx::k::part3_scatter_kernel<plx::k::StatProg<13>, 0, 2, 1, false, false>(plx::fused::Shape, plx::fused::Args, plx::fused::PartPlan2, plx::k::ScatterParams2)", "part3_scatter[#13,h,t2,p1]"),
    ("void plx::k::part3_scatter_kernel<plx::k::StatProg<13>, 0, 2, 1, false, true>(plx::fused::Shape, plx::fused::Args, plx::fused::PartPlan2, plx::k::ScatterParams2)", "part3_scatter[#13,h,t2,p1]"),
    ("void plx::k::part3_scatter_kernel<plx::k::StatProg<4>, 1, 3, 2, true, true>(plx::fused::Shape, plx::fused::Args, plx::fused::PartPlan2, plx::k::ScatterParams2)", "part3_scatter[#4,d,t3,p2,hot]"),
    ("void plx::k::part2_agg_kernel<plx::k::StatProg<13>, 0, 1>(plx::k::PartPlan2, plx::k::AggParams2)", "part_agg_lds[#13,h,p1]"),
    ("plx::k::canonicalise_chains_kernel(plx::fused::JoinAggTable, plx::fused::RepCols, unsigned int, unsigned int*)", "join_chain_representatives"),
    ("plx::k::rows_agg_compact_kernel(unsigned long long const*, long, int, int, unsigned long long*, unsigned int*, unsigned long long*)", "table_compact"),
    ("plx_jit_part3_scatter_21_0a1b2c3d", "part3_scatter[jit,h,t2,p0]"), ("plx_jit_part3_scatter_60_0a1b2c3d", "part3_scatter[jit,d,t1,p1,hot]"), ("plx_jit_part3_agg_85_deadbeef", "part_agg_lds[jit,h,p1]"),
    # (round 6: run-time compiled plain scans resolve to the tracer's name for a scan without an AOT kernel -- the JIT-shape bench extra's traffic is looked up by it)
    ("plx_jit_RegAggSink_0_0a1b2c3d", "fused_scan_regagg_generic"), ("plx_jit_LdsAggSink_1_0a1b2c3d", "fused_scan_ldsagg_generic"), ("plx_jit_BallotSink_91_0a1b2c3d", "fused_scan_ballots[jit]"),
    ("void plx::k::compact_by_ballots_kernel<8, 3>(plx::k::CompactCols, unsigned long long const*, unsigned long long const*, long, unsigned int*)", "filter_compact_cols"),
    ("void plx::k::fused_scan_kernel<plx::k::StatProg<12>, plx::k::DirectHitsSink>(plx::fused::Shape, plx::fused::Args, plx::k::DirectHitsSink::Params)", "fused_scan_direct_hits_static#12"),
    ("plx_jit_part3_scatter_94_0badf00d", "part3_scatter[jit,h,t2,p4]"), ("plx_jit_part3_scatter_108_0badf00d", "part3_scatter[jit,d,t4,p4,hot]"),
    ("plx_jit_part3_agg_109_0badf00d", "part_agg_lds[jit,h,p4]"), ("plx_jit_part3_agg_110_0badf00d", "part_agg_lds[jit,d,p4]"),
    ("plx::k::join_bin_kernel(plx::fused::DirectJoinTable, plx::fused::JoinAggTable, HIP_vector_type<unsigned long long, 2u>*, unsigned int*)", "join_bin_windows"),
    ("plx::k::join_fill_kernel(HIP_vector_type<unsigned long long, 2u> const*, unsigned int const*, unsigned long long const*, plx::fused::JoinAggTable, unsigned long long*, unsigned int*)", "join_fill_lds"),
    ("plx::k::cells_agg_compact_kernel(unsigned long long const*, unsigned int const*, unsigned long long const*, long, int, int, unsigned long long*, unsigned long long*, unsigned int*, unsigned long long*)", "table_compact"),
    ("plx::join::join_match_kernel(plx::join::KeyCol, unsigned int const*, long, plx::join::PairTable, int, unsigned int*, unsigned int*, unsigned long long*)", "join_match"),
    ("plx_jit_UnknownSink_7_0a1b2c3d", None),
    ("__amd_rocclr_fillBufferAligned", None),
]


def test_kernel_symbols_map_to_the_tracer_names():
    for symbol, name in CASES:
        assert pmc.scope_of(symbol) == name, (symbol, pmc.scope_of(symbol), name)


ROUND = "r06"


def test_committed_summaries_are_keyed_by_symbol_and_cover_every_kernel_that_matters():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", ROUND, "*_pmc.json")))
    assert len(files) >= 12
    for f in files:
        d = json.load(open(f))
        assert d.get("keyed_by") == "kernel symbol", f
        wl = os.path.basename(f)[:-len("_pmc.json")]
        # (a) by symbol: the kernel that takes most of the workload's time resolves to a name with a counter figure
        stats = os.path.join(ROOT, "profiles", ROUND, wl + "_kernel_stats.csv")
        # (input preparation is not the workload: generators, the gathers that shuffle q3s' tables -- but the gather IS the gather workload -- and index ramps)
        rows = [r for r in csv.DictReader(open(stats)) if ("plx::" in r["Name"] or r["Name"].startswith("plx_jit_")) and "datagen" not in r["Name"] and "iota_kernel" not in r["Name"]
                and (wl == "gather" or "gather_kernel" not in r["Name"])]
        rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
        top = pmc.scope_of(rows[0]["Name"])
        assert top is not None and top in d["kernels"], (f, rows[0]["Name"][:80], top, list(d["kernels"])[:6])
        assert d["kernels"][top]["hbm_bytes_per_launch"] > 0
        # (b) by tracer name: every kernel of the same session's bench line with >= 3 % of the step's kernel time has a counter figure under that very name
        line = json.load(open(os.path.join(ROOT, "profiles", ROUND, wl + "_bench_line_same_session.json")))
        total = sum(v["avg_us"] * v["launches"] for v in line["kernels"].values())
        for name, v in line["kernels"].items():
            if v["avg_us"] * v["launches"] >= 0.03 * total:
                assert name in d["kernels"], (wl, name, sorted(d["kernels"]))
