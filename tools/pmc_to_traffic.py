#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; tools/capture_profiles.sh pmc layout) into profiles/*_pmc_traffic.json:
HBM bytes per launch per kernel family = (2*FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the family's launches
(KB units; gfx950 FETCH_SIZE counts half of a wide coalesced read -- MI355X_MICROARCH.md, HBM section).
usage: pmc_to_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/> <out.json> [per-kernel csv]"""
import collections, csv, glob, json, sys

FAMILIES = [("fps_kernel", "fps"), ("query_ball_point_kernel", "ball_query+group"), ("group_point_kernel", "ball_query+group"),
            ("group_xyz_kernel", "ball_query+group"), ("sa1_fused_kernel", "shared_mlp_fused_sa"), ("sa2_fused_kernel", "shared_mlp_fused_sa"),
            ("sa3_chain_kernel", "shared_mlp_conv1x1"), ("fp1_chain_kernel", "shared_mlp_conv1x1"), ("fp2_chain_kernel", "shared_mlp_conv1x1"), ("fp_init_kernel", "shared_mlp_conv1x1"),
            ("conv1x1_kernel", "shared_mlp_conv1x1"), ("conv1x1_few_rows_kernel", "shared_mlp_conv1x1"), ("conv_packed_kernel", "shared_mlp_conv1x1"), ("conv_rowtile_kernel", "shared_mlp_conv1x1"),
            ("three_nn_kernel", "three_nn+interpolate"), ("three_weights_kernel", "three_nn+interpolate"),
            ("three_interpolate_kernel", "three_nn+interpolate"), ("fp_concat_kernel", "three_nn+interpolate"),
            ("group_xyz_multi_kernel", "ball_query+group"), ("group_point_multi_kernel", "ball_query+group"), ("query_ball_lanes_kernel", "ball_query+group"),
            ("mlp_chain_wave_kernel", "shared_mlp_chain_tail"), ("mlp_chain1_kernel", "shared_mlp_chain_tail"), ("mlp_chain_kernel", "shared_mlp_chain_tail"),
            ("head_act_kernel", "head_activations")]


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return None


def main(src, out, per_kernel=None):
    per = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> values
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("%s/%s/*counter_collection.csv" % (src, c))
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == c:
                per[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
    fam_bytes = collections.defaultdict(list)
    rows = [("kernel", "launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "hbm_bytes_per_launch")]
    for k, v in sorted(per.items()):
        fe = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])); wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
        b = (2 * fe + wr) * 1024
        rows.append((k[:120], len(v["FETCH_SIZE"]), round(fe, 1), round(wr, 1), round(b)))
        fam = family(k)
        if fam:
            fam_bytes[fam] += [b] * len(v["FETCH_SIZE"])
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 2 --warmup 1 --slots 1 --no-graph`; "
                     "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (KB units; gfx950 FETCH_SIZE counts half of a wide coalesced read, "
                     "MI355X_MICROARCH.md HBM section); averaged per launch over the kernel family (tools/pmc_to_traffic.py)",
           "hbm_bytes_per_launch": {k: round(sum(v) / len(v)) for k, v in fam_bytes.items()},
           "family_launches_in_the_pass": {k: len(v) for k, v in fam_bytes.items()},
           "hbm_bytes_per_family_in_the_pass": {k: round(sum(v)) for k, v in fam_bytes.items()}}
    try:                                   # keep hand-collected op-level entries (tools/capture_profiles.sh ops sections) across regenerations
        old = json.load(open(out))
        res.update({k: v for k, v in old.items() if k.startswith("ops_")})
    except (OSError, ValueError):
        pass
    json.dump(res, open(out, "w"), indent=1)
    if per_kernel:
        csv.writer(open(per_kernel, "w", newline="")).writerows(rows)
    print(json.dumps(res["hbm_bytes_per_launch"], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
