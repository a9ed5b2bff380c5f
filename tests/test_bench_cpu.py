"""bench.py's host-side bookkeeping (no GPU): the per-call work formulas know every entry point the pipeline issues, and a PMC
figure is only reported while the kernel sources it was measured on are unchanged."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_kernel_work_of_the_grouped_entry_points():
    b = _bench()
    # two networks per launch = the sum of the plain calls
    fam, by, fl = b.kernel_work("ancsh_sa_module_fused_grouped", (2, 32, 1024, 512, 64, 0, 64, 64, 128))
    fam1, by1, fl1 = b.kernel_work("ancsh_sa_module_fused", (32, 1024, 512, 64, 0, 64, 64, 128))
    assert fam == fam1 == "shared_mlp_fused_sa" and fl == 2 * fl1 and by1 < by < 2 * by1          # the geometry is read once
    fam, by, fl = b.kernel_work("ancsh_sa_module_fused_partial_grouped", (2, 32, 512, 128, 64, 128, 128, 256))
    fam1, by1, fl1 = b.kernel_work("ancsh_sa_module_fused_partial", (32, 512, 128, 64, 128, 128, 256))
    assert fam == "shared_mlp_fused_sa" and fl == 2 * fl1
    g = b.kernel_work("ancsh_conv1x1_packed_grouped", (2, 4096, 256, 256, 0, 256, 0, 0, 0, 0, 1, 0, 256, 0, 0, 0))
    p = b.kernel_work("ancsh_conv1x1_packed", (4096, 256, 256, 0, 256, 0, 0, 0, 0, 1, 0, 256, 0, 0, 0))
    assert g[0] == p[0] == "shared_mlp_conv1x1" and g[2] == 2 * p[2] and g[1] == 2 * p[1]
    assert b.kernel_work("ancsh_ransac_single_ex", (96, 0, 0, 0, 0.1, 10000))[0] == "pose_ransac_single"
    assert b.kernel_work("ancsh_three_nn_weights", (32, 1024, 512))[0] == "three_nn+interpolate"
    assert b.kernel_work("ancsh_fp_interpolate_concat_ex", (64, 512, 128, 1024, 0, 0, 0, 0, 3, 0, 132, 32, 32))[0] == "three_nn+interpolate"
    assert b.kernel_work("ancsh_conv1x1_grouped", (2, 32, 1024, 256))[0] == "fp_partial_product(valu)"
    # both networks' tail chains in one launch (round 4) = the ANCSH program's flops + the NPCS program's
    b.CHAIN_FLOPS[(32768, 11)], b.CHAIN_FLOPS[(32768, 8)] = b.chain_flops(32768, 3, True), b.chain_flops(32768, 3, False)
    fam, _by, fl = b.kernel_work("ancsh_mlp_chain_grouped", (2, 32768, 131, 0, 132))
    assert fam == "shared_mlp_chain_tail" and fl == b.chain_flops(32768, 3, True) + b.chain_flops(32768, 3, False)
    assert b.kernel_work("ancsh_mlp_chain_grouped", (1, 32768, 131, 0, 132))[2] == b.chain_flops(32768, 3, True)


def test_split16_entry_points_are_priced_against_the_16_bit_peak():
    """The split-16 experiment's launches: the f32-EQUIVALENT flops of their f32 counterparts, in families of their own whose `achieved`
    is the 16-bit products they execute (6 per f32 product for Bf16x3, 3 for F16x2) against the dense 16-bit matrix peak."""
    b = _bench()
    a = (2, 32, 1024, 512, 64, 0, 64, 64, 128)
    f32 = b.kernel_work("ancsh_sa_module_fused_grouped", a)
    for scheme, prod in (("bf16x3", 6), ("f16x2", 3)):
        fam, by, fl = b.kernel_work("ancsh_sa_module_fused_%s_grouped" % scheme, a)
        assert fam == "shared_mlp_fused_sa [%s]" % scheme and (by, fl) == f32[1:]
        assert b.kernel_work("ancsh_sa_module_fused_partial_%s_grouped" % scheme, (2, 32, 512, 128, 64, 128, 128, 256))[0] == fam
        assert b.kernel_work("ancsh_fp2_chain_grouped_%s" % scheme, (2, 32, 128, 512, 256, 128, 256, 128))[0] == "shared_mlp_conv1x1 [%s]" % scheme
        assert b.kernel_work("ancsh_sa_pack_weights_%s" % scheme, (64, 64))[2] == 0.0
        roof = b.roofline_from_profile([("ancsh_sa_module_fused_%s_grouped" % scheme, a, 0.2)], 1)[fam]
        eq = fl / 0.2e-3 / 1e12
        assert roof["bound"] == "mfma" and roof["peak"] == 2500.0 and roof["products_per_f32_product"] == prod
        assert abs(roof["f32_equivalent_TFLOPs"] - eq) < 0.01 and abs(roof["achieved"] - prod * eq) < 0.1 and abs(roof["frac"] - prod * eq / 2500.0) < 1e-3
    b.CHAIN_FLOPS[(32768, 11)], b.CHAIN_FLOPS[(32768, 8)] = b.chain_flops(32768, 3, True), b.chain_flops(32768, 3, False)
    fam, _by, fl = b.kernel_work("ancsh_mlp_chain_grouped_fp_f16x2", (2, 32, 1024, 512, 128))
    assert fam == "shared_mlp_chain_tail [f16x2]" and fl == b.chain_flops(32768, 3, True) + b.chain_flops(32768, 3, False)


def test_pmc_figures_go_null_when_the_kernel_sources_changed(tmp_path, monkeypatch):
    b = _bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    now = b.source_digests()
    assert "sa_fused.hip" in now and "common.h" in now and all(len(v) == 16 for v in now.values())
    good = {"commit": "abc1234", "source_digests": now, "hbm_bytes_per_launch": {"shared_mlp_fused_sa": 123, "fps": 7},
            "ops_ball_query+group_hbm_bytes_per_batch": 99}
    (prof / "r09_pmc_traffic.json").write_text(json.dumps(good))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    assert b.pmc_traffic() == {"shared_mlp_fused_sa": 123, "fps": 7}
    assert b.pmc_entry("ops_ball_query+group_hbm_bytes_per_batch", ("grouping.hip",)) == 99
    assert b.pmc_provenance() == {"file": "profiles/r09_pmc_traffic.json", "commit": "abc1234", "stale_sources": []}
    stale = dict(good, source_digests=dict(now, **{"sa_fused.hip": "0" * 16}))
    (prof / "r09_pmc_traffic.json").write_text(json.dumps(stale))
    t = b.pmc_traffic()
    assert t["shared_mlp_fused_sa"] is None and t["fps"] == 7                      # only the family whose source changed
    assert b.pmc_provenance()["stale_sources"] == ["sa_fused.hip"]
    other = dict(good, source_digests=dict(now, **{"metrics.hip": "0" * 16, "error.cpp": "1" * 16}))     # no kernel of the step in either
    (prof / "r09_pmc_traffic.json").write_text(json.dumps(other))
    assert b.pmc_traffic() == {"shared_mlp_fused_sa": 123, "fps": 7}
    assert b.pmc_provenance()["stale_sources"] == [] and b.pmc_provenance()["other_changed_sources"] == ["error.cpp", "metrics.hip"]
    (prof / "r09_pmc_traffic.json").write_text(json.dumps({k: v for k, v in good.items() if k != "source_digests"}))
    assert b.pmc_traffic() == {"shared_mlp_fused_sa": None, "fps": None}           # an unstamped file proves nothing


def test_final_line_fits_the_drivers_capture():
    """The contract line stays below 4 KB whatever the legs carry (round 5's 20 KB line was cut by the driver's bounded stdout
    capture: `BENCH_r05.json.parsed` null), and still holds `roofline` and `cpu_baseline`."""
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "r05_bench_driver_cmd_steps20_warmup5.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    c = b.compact_line(full, "bench_detail.json")
    s = json.dumps(c)
    assert len(s) < b.FINAL_LINE_MAX == 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype"):
        assert c[k] == full[k]
    assert c["data"] == "synthetic" and c["config"]["workload"].startswith("configs[2]")
    r = c["roofline"]
    assert r["kernel"] == "shared_mlp_fused_sa" and r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    cb = c["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 13 and cb["kind"] == "port" and cb["sample"] and cb["single_core"]["cores"] == 1
    assert len(c["value_configs"]) == 3 and c["roofline_ops"]["ball_query+group"]["frac"] == full["roofline_ops"]["ball_query+group"]["frac"]
    # eight ranks and absurdly long strings still fit: optional legs are shed before the contract fields
    fat = dict(full, ranks=[dict(full["ranks"][0], rank=i, pid=1000 + i) for i in range(8)])
    fat["cpu_baseline"] = dict(full["cpu_baseline"], sample="x" * 5000)
    fat["value_configs"] = full["value_configs"] * 6
    c2 = b.compact_line(fat, "bench_detail.json")
    assert len(json.dumps(c2)) < 4096 and "roofline" in c2 and "cpu_baseline" in c2 and c2["value"] == full["value"]
    # a split-16 leg's own roofline rides along in a few fields
    leg = {"value": 31000.0, "ms_per_step": 1.03, "steps": 20, "speedup_vs_value": 1.5, "roofline_all": {"x": {"frac": 1}},
           "roofline": {"kernel": "shared_mlp_fused_sa [f16x2]", "bound": "mfma", "achieved": 760.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.304,
                        "f32_equivalent_TFLOPs": 253.3, "timing": "y" * 900}}
    c3 = b.compact_line(dict(full, value_f16x2=leg), "bench_detail.json")
    assert c3["value_f16x2"]["roofline"] == {"kernel": "shared_mlp_fused_sa [f16x2]", "frac": 0.304, "achieved": 760.0, "f32_equivalent_TFLOPs": 253.3,
                                             "peak_TFLOPs": 2500.0} and len(json.dumps(c3)) < 4096


def test_emit_prints_the_contract_line_last(tmp_path, monkeypatch, capsys):
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "r05_bench_driver_cmd_steps20_warmup5.json")) as f:
        full = json.load(f)
    monkeypatch.setenv("ANCSH_BENCH_DETAIL", str(tmp_path / "d.json"))
    b.print_last(b.emit(full))
    cap = capsys.readouterr()
    out = [l for l in cap.out.splitlines() if l.strip()]
    assert len(out) == 1                                                           # stdout: the contract line and nothing else
    assert json.loads([l for l in cap.err.splitlines() if l.startswith("{")][-1])["bench_detail"] == full
    last = json.loads(out[-1])
    assert len(out[-1]) < 4096 and last["roofline"]["frac"] == full["roofline"]["frac"] and last["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert json.load(open(tmp_path / "d.json")) == full and last["detail"] == str(tmp_path / "d.json")


def test_step_account_splits_by_pipe_work(tmp_path):
    """tools/step_account.py --counters: co-running dispatches share an interval in proportion to their pipe-work rates (SQ counters),
    not to the SIMDs they could occupy -- a matrix kernel next to a 4096-wave, nearly idle kernel gets (almost) the whole interval,
    and no family's pipe_ms falls below its work at 100 % issue (round 5's table put fused SA at 163 TFLOP/s)."""
    import csv
    import importlib.util
    spec = importlib.util.spec_from_file_location("step_account", os.path.join(ROOT, "tools", "step_account.py"))
    sa = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sa)
    tr, ct = tmp_path / "t.csv", tmp_path / "c.csv"
    rows, t = [], 0
    for step in range(12):       # per step: the marker (LM, 64 waves), then an SA launch [t, t+500us) with a light 4096-wave kernel inside it
        rows.append(("ancsh::pose::ransac_joint_lm_kernel(int)", t, t + 10_000, 64, 64))
        rows.append(("void ancsh::sa1_fused_kernel<64, 64, 128>(int)", t + 10_000, t + 510_000, 256, 32768 * 64))
        rows.append(("ancsh::query_ball_kernel(int)", t + 110_000, t + 210_000, 256, 4096 * 64))
        t += 600_000
    with open(tr, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size_X", "Grid_Size_X"])
        for name, s, e, wg, grid in rows:
            w.writerow(["KERNEL_DISPATCH", name, s, e, wg, grid])
    chip = 1024 * 2.4            # SIMD-cycles per ns
    with open(ct, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "family", "launches", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES"])
        w.writerow(["ancsh::sa1_fused_kernel(int)", "fused SA (MFMA)", 5, 0.8 * chip * 500_000 / 64, 0.8 * chip * 500_000 / 64, 0.8 * chip * 500_000])
        w.writerow(["ancsh::query_ball_kernel(int)", "ball query", 5, 0, 0.02 * chip * 100_000 / 4, 0])
        w.writerow(["ancsh::pose::ransac_joint_lm_kernel(int)", "pose stage B: LM fits", 5, 0, 0.05 * chip * 10_000 / 4, 0])
    res = sa.account(sa.load(str(tr)), 0, 0.0, work=sa.load_counters(str(ct)))
    n = res["steps"]
    per = lambda d, fam: d[fam] / n * 1e-6
    pipe = res["pipe"]
    assert abs(per(pipe, "fused SA (MFMA)") - (0.400 + 0.100 * 0.8 / 0.82)) < 2e-3            # 400 us alone + its share of the shared 100 us
    assert abs(per(pipe, "ball query") - 0.100 * 0.02 / 0.82) < 1e-3
    floor = res["pwork"]["fused SA (MFMA)"] / (1024 * 2.4) / n * 1e-6
    assert abs(floor - 0.400) < 2e-3 and per(pipe, "fused SA (MFMA)") >= floor
    assert per(res["attributed"], "fused SA (MFMA)") < 0.46                                   # the equal-share view charges the shared interval 50 / 50
    import io
    buf = io.StringIO()
    sa.report(res, buf)
    assert "pipe_ms" in buf.getvalue() and "floor_ms" in buf.getvalue()


def test_a_leg_hands_its_full_record_to_the_parent_on_stdout(capsys):
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "r05_bench_driver_cmd_steps20_warmup5.json")) as f:
        full = json.load(f)
    b.print_last(b.emit(full, sidecar=False))
    out = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(out) == 2 and json.loads(out[0])["bench_detail"] == full and "detail" not in json.loads(out[1])
