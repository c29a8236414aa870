#!/usr/bin/env python
"""rocprofv3 counter-collection CSVs (FETCH_SIZE and WRITE_SIZE passes of tools/pmc_all.sh) -> profiles/<round>/<workload>_pmc.json:
HBM bytes per (median) launch of every library kernel, keyed by the name bench.py's HIP-event tracer uses for it (ProfileScope), with the
gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as 64 B: x2 for wide coalesced reads) and a write
calibration on the library's own generator kernel (datagen_uniform writes exactly 8 B per row).
usage: pmc_summarise.py <workload> <fetch counter csv> <write counter csv> <out json> [steps]"""
import collections
import csv
import json
import re
import sys

SCOPES = [  # (regex on the demangled kernel name, ProfileScope name in bench.py)
    (r"fused_scan_kernel<.*DirectHitsSink", "fused_scan_direct_hits_static"), (r"fused_scan_kernel<.*BallotSink", "fused_scan_ballots_static"),
    (r"fused_scan_kernel<.*LdsAggSink", "fused_scan_ldsagg_static"), (r"fused_scan_kernel<.*RegAggSink", "fused_scan_regagg_static"),
    (r"fused_scan_kernel<.*DirectProbeAggSink", "fused_scan_direct_probe_agg_static"), (r"fused_scan_kernel<.*DirectBuildSink", "fused_scan_direct_build_static"),
    (r"fused_scan_kernel<.*BitmapBuildSink", "fused_scan_bitmap_build_static"), (r"fused_scan_kernel<.*HashAggSink", "fused_scan_hashagg"),
    (r"fused_scan_kernel<.*ProbeAggSink", "fused_scan_probe_agg_static"), (r"fused_scan_kernel<.*JoinBuildSink", "fused_scan_join_build_static"),
    (r"part2_scatter_kernel", "part2_scatter"), (r"part2_agg_kernel", "part2_agg_lds"), (r"chunk_hist_kernel|chunk_place_kernel", "part2_chunk_sort"),
    (r"part_scatter_kernel", "part_scatter"), (r"part_agg_kernel", "part_agg_lds"), (r"part_count_kernel", "part_count"),
    (r"direct_popc_kernel|direct_word_rank_kernel", "direct_rank"), (r"direct_pairs_compact_kernel|hash_compact_kernel|join_agg_compact_kernel", "table_compact"),
    (r"probe_pass_kernel", "probe_pass_lds"), (r"probe_hits_compact_kernel", "probe_hits_compact"), (r"touch_filter_kernel", "touch_filter"),
    (r"scan_block_kernel|scan_add_kernel|scan_", "exclusive_scan"), (r"gather_multi_kernel", "gather_multi"), (r"gather_multi_kernel", "gather_multi"), (r"gather_kernel", "gather_u32"), (r"finalize_batch_kernel", "finalize_batch"),
    (r"datagen_uniform_kernel<long", "datagen_uniform_i64"), (r"datagen_", "datagen_other"), (r"hot_candidates_kernel|hot_emit_kernel", "hot_keys"),
    (r"strgroup_scatter_kernel", "strgroup_scatter"), (r"strgroup_agg_kernel", "strgroup_agg_lds"),
    (r"canonicalise_chains_kernel", "join_chain_representatives"), (r"chains_count_kernel|chains_emit_kernel", "table_compact"), (r"rows_agg_compact_kernel|wide_compact_kernel", "table_compact"),
    (r"fused_scan_kernel<.*WideAggSink", "fused_scan_wideagg"),
    (r"compact_by_ballots_kernel", "filter_compact_cols"), (r"join_match_kernel", "join_match"), (r"join_pairs_emit_kernel", "join_pairs_emit"), (r"filter_rowids_kernel|ballots_to_rowids_kernel", "filter_rowids"), (r"direct_slot_rows_kernel", "direct_slot_rows"),
    (r"strview_stamp_kernel", "strview_stamp_nulls"),
    (r"join_bin_kernel", "join_bin_windows"), (r"join_fill_kernel", "join_fill_lds"), (r"cells_agg_compact_kernel", "table_compact"),
    (r"filter_kernel<", "filter_compact"), (r"tile_count_kernel", "filter_tile_count"), (r"ballots_to_mask_kernel", "ballots_to_mask"),
    (r"init_acc_kernel|fill_u64_kernel", "table_init"), (r"strview_encode_kernel", "strview_dict_encode"), (r"strdict_", "strdict_materialise"),
]


def scope_of(kernel: str):
    """The name the library's HIP-event profile gives a launch of this kernel SYMBOL: AOT instantiations carry their template
    arguments (kernels_fused.hip scope_name, kernels_partition.hip partitioned_agg2), so bench.py only ever pairs a counter figure
    with the very instantiation it timed."""
    # run-time compiled kernels carry their kind in the symbol (jit.cpp kernel_symbol: plx_jit_<kind>_<sink number>_<shape hash>)
    m = re.match(r"plx_jit_part3_scatter_(\d+)_", kernel)
    if m and int(m.group(1)) >= 111:     # ... with the value as a 48-bit offset (PART3_SCATTER_PAIRV = 111: + (tiles - 1) + 4 * hot; hash mode)
        v = int(m.group(1)) - 111
        return f"part3_scatter[jit,h,t{1 + (v & 3)},p5{',hot' if (v >> 2) & 1 else ''}]"
    if m and int(m.group(1)) >= 93:      # two rows a record (jit.hpp PART3_SCATTER_PAIR = 93: + (tiles - 1) + 4 * hot + 8 * mode)
        v = int(m.group(1)) - 93
        return f"part3_scatter[jit,{'d' if v >> 3 else 'h'},t{1 + (v & 3)},p4{',hot' if (v >> 2) & 1 else ''}]"
    if m:
        v = int(m.group(1)) - 19
        mode, tiles, pack, hot = v & 1, 1 + ((v >> 1) & 3), (v >> 3) & 3, v >> 5
        return f"{'probe_scatter' if pack == 3 else 'part3_scatter'}[jit,{'d' if mode else 'h'},t{tiles},p{pack}{',hot' if hot else ''}]"
    m = re.match(r"plx_jit_part3_agg_(\d+)_", kernel)
    if m and int(m.group(1)) == 119:     # PART3_AGG_PAIRV
        return "part_agg_lds[jit,h,p5]"
    if m and int(m.group(1)) >= 109:     # PART3_AGG_PAIR = 109 (hash), 110 (direct)
        return f"part_agg_lds[jit,{'d' if int(m.group(1)) == 110 else 'h'},p4]"
    if m:
        v = int(m.group(1)) - 83
        return f"part_agg_lds[jit,{'d' if v & 1 else 'h'},p{v >> 1}]"
    # run-time compiled plain scan sinks: plx_jit_<Sink>_<kind>_<hash> -> the name the tracer gives a scan without an AOT kernel (kernels_fused.hip scope_name)
    m = re.match(r"plx_jit_(\w+Sink)_\d+_", kernel)
    if m:
        return {"LdsAggSink": "fused_scan_ldsagg_generic", "RegAggSink": "fused_scan_regagg_generic", "BallotSink": "fused_scan_ballots[jit]", "JoinBuildSink": "fused_scan_join_build",
                "ProbeAggSink": "fused_scan_probe_agg", "DirectBuildSink": "fused_scan_direct_build", "DirectProbeAggSink": "fused_scan_direct_probe_agg",
                "BitmapBuildSink": "fused_scan_bitmap_build", "DirectHitsSink": "fused_scan_direct_hits", "HashAggSink": "fused_scan_hashagg", "DenseAggSink": "fused_scan_denseagg", "WideAggSink": "fused_scan_wideagg"}.get(m.group(1))
    m = re.search(r"fused_scan_kernel<.*StatProg<(\d+)>", kernel)
    sid = m.group(1) if m else None
    m = re.search(r"part3_scatter_kernel<.*StatProg<(\d+)>\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*(?:,\s*(true|false)\s*)?(?:,\s*(?:true|false)\s*)?>", kernel)      # (the last flag: the per-row check of narrowed values compiled in)
    if m:      # pack 3 (row-id records) is the probe side's scatter of the partitioned join probe: the library's tracer calls it probe_scatter
        hot = ",hot" if m.group(5) == "true" else ""
        return f"{'probe_scatter' if m.group(4) == '3' else 'part3_scatter'}[#{m.group(1)},{'d' if m.group(2) == '1' else 'h'},t{m.group(3)},p{m.group(4)}{hot}]"
    m = re.search(r"part2_scatter_kernel<.*StatProg<(\d+)>\s*,\s*(\d+)\s*,\s*(\d+)\s*>", kernel)
    if m:
        return f"part2_scatter[#{m.group(1)},{'d' if m.group(2) == '1' else 'h'},t{m.group(3)}]"
    m = re.search(r"part2_agg_kernel<.*StatProg<(\d+)>\s*,\s*(\d+)\s*,\s*(\d+)\s*>", kernel)
    if m:
        return f"part_agg_lds[#{m.group(1)},{'d' if m.group(2) == '1' else 'h'},p{m.group(3)}]"
    for rx, name in SCOPES:
        if re.search(rx, kernel):
            return f"{name}#{sid}" if sid is not None and name.endswith("_static") else name
    return None


def read(path, counter):
    """kernel symbol -> [launches, launches x MEDIAN counter value per launch].  The median, not the mean: a process also launches a kernel outside the steps
    it measures (the q3s workload PREPARES its input with eight SF100-sized gathers; the same gather kernel then runs sixty times on a few million rows inside
    the steps) -- the typical launch is what a step's traffic is made of."""
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    agg = {}
    for k, v in vals.items():
        v.sort()
        med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
        agg[k] = [len(v), med * len(v)]
    return agg


def main():
    wl, fetch_csv, write_csv, out = sys.argv[1:5]
    fetch, write = read(fetch_csv, "FETCH_SIZE"), read(write_csv, "WRITE_SIZE")
    kernels = {}
    for src, field in ((fetch, "fetch_KB"), (write, "write_KB")):
        for k, (n, tot) in src.items():
            sc = scope_of(k)
            if sc is None:
                continue
            e = kernels.setdefault(sc, {"fetch_KB": 0.0, "write_KB": 0.0, "launches": 0, "kernel_names": []})
            e[field] += tot
            if field == "fetch_KB":
                e["launches"] += n
            if k[:100] not in e["kernel_names"]:
                e["kernel_names"].append(k[:100])
    res = {}
    for sc, e in kernels.items():
        n = max(e["launches"], 1)
        if sc == "part2_chunk_sort":
            n = max(n // 2, 1)          # two kernels per pass of the chunk sort
        if sc == "direct_rank":
            n = max(n // 2, 1)
        if sc == "table_compact" and any("chains_count_kernel" in k for k in e["kernel_names"]):
            n = max(n // 2, 1)          # the multi-value compaction is two kernels (count, emit) under one tracer scope
        f, w = e["fetch_KB"] / n, e["write_KB"] / n
        res[sc] = {"fetch_KB": round(f, 1), "write_KB": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024), "launches_seen": e["launches"], "kernel_names": e["kernel_names"]}
    cal = res.get("datagen_uniform_i64")
    import time
    doc = {"workload": wl, "keyed_by": "kernel symbol", "collected": time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()),
           "command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) --kernel-trace -- python bench.py --workload {wl} --steps 3 --warmup 1 --no-extras --no-cpu  (tools/pmc_all.sh)",
           "correction": "gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported, calibrated on datagen_uniform_i64 "
                         "(the library's generator writes exactly 8 B per row): see `calibration`",
           "calibration": cal, "kernels": res}
    json.dump(doc, open(out, "w"), indent=1)
    for sc, e in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
        print(f"{wl:6s} {sc:38s} hbm {e['hbm_bytes_per_launch'] / 1e9:8.3f} GB  (fetch {e['fetch_KB'] * 1024 * 2 / 1e9:7.3f}  write {e['write_KB'] * 1024 / 1e9:7.3f})")


if __name__ == "__main__":
    main()
