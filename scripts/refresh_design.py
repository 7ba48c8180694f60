"""Rewrites the measured figures of DESIGN.md (the kernel table of section 4, the tables of sections 5 and 8, the headline of section 0)
from the files under profiles/ they are quoted from, so that the document cannot drift from its evidence.
Usage: python scripts/refresh_design.py   (after copying a run's artefacts to profiles/r06_final_*)"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def line(f):
    return json.loads([ln for ln in open(os.path.join(P, f)) if ln.strip().startswith("{")][0])


def main():
    b, d = line("r06_final_bench.json"), line("r06_final_bench_driver.json")
    c1, c1n = line("r06_final_bench_c1.json"), line("r06_final_bench_c1_noprofile.json")
    c4, c5, mp = line("r06_final_bench_c4.json"), line("r06_final_bench_c5.json"), line("r06_final_bench_mapping.json")
    ad = json.load(open(os.path.join(P, "r06_adapter_latency.json")))
    adc = json.load(open(os.path.join(P, "r06_adapter_latency_copy_frames.json")))
    tr = json.load(open(os.path.join(P, "r06_final_pmc_traffic.json")))
    t, git = tr["kernels"], tr["meta"].get("git", "unknown")
    rc, pl = line("r06_final_rccl_single_rank.json"), line("r06_final_plain_single_rank.json")
    tw, tt = line("r06_final_two_ranks_one_gpu.json"), line("r06_final_torchrun_two_ranks_one_gpu.json")
    sl = json.load(open(os.path.join(P, "r06_small_latency.json")))
    ntests = re.search(r"(\d+) passed", open(os.path.join(P, "r06_final_pytest_gpu.log")).read()).group(1)
    k = b["kernels"]
    step_k = ["k_build_AS", "k_chol_left", "k_fwdsub_lds", "k_syrk", "k_search_mfma", "k_feature_prediction", "k_finalize", "k_predict",
              "k_search_score", "k_select"]
    tot = sum(t[x]["hbm_bytes"] for x in step_k) / 1e9
    work = b["work_per_step"]
    fwd_frac = work["sum_nmm"] / (k["k_fwdsub_lds"]["ms_per_step"] * 1e-3) / 1e12 / 78.6
    rest = sum(k[x]["ms_per_step"] for x in ("k_predict", "k_feature_prediction", "k_select", "k_finalize", "k_search_score"))
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()

    s = re.sub(r"Headline \(driver protocol.*?smoke 5\.4e-15 / 2\.0e-14\.",
               "Headline (driver protocol `--steps 20 --warmup 5`): **%d k sequence-frames/s, %.3f ms per step** of 1024 sequences x 100 features in the\n"
               "committed final run (library %s; before the placement of section 3: 654-697 k over the round's boxes and protocols; r05: 673 k / 1.521).\n"
               "GPU tests %s passed; smoke 5.4e-15 / 2.0e-14." % (round(d["value"] / 1e3), d["ms_per_step"], git, ntests), s, flags=re.S)

    def cell(pattern, new):
        nonlocal s
        m = re.search(r"(\| %s[^\n]*\| )([^|\n]*)( \|\n)" % pattern, s)
        assert m, pattern
        s = s[:m.start(2)] + new + s[m.end(2):]

    cell(re.escape("`k_search_mfma` + `k_search_score`"), "%.4f ms, %.3f of 8 TB/s; MFMA floor 15.8 us" % (k["k_search_mfma"]["ms_per_step"], b["roofline"]["search"]["frac"]))
    cell(re.escape("`k_build_AS`"), "%.3f ms (on the allocation `sl2_create` chose, section 3; 0.335-0.357 on a slow one)" % k["k_build_AS"]["ms_per_step"])
    cell(re.escape("`k_chol_left`"), "%.3f ms" % k["k_chol_left"]["ms_per_step"])
    cell(re.escape("`k_fwdsub_lds`"), "%.3f ms = %.2f of 78.6 TFLOP/s" % (k["k_fwdsub_lds"]["ms_per_step"], fwd_frac))
    cell(re.escape("`k_syrk` (the dominant kernel"), "%.3f ms = **%.3f** of 78.6 TFLOP/s" % (k["k_syrk"]["ms_per_step"], b["roofline"]["frac"]))
    cell(re.escape("`k_predict`, `k_feature_prediction`, `k_select`, `k_finalize`"), "%.3f ms together (with `k_search_score`)" % rest)
    s = re.sub(r"The step moves [0-9.]+ GB \([^)]*\)", "The step moves %.2f GB (%s)" % (
        tot, ", ".join("`%s` %.2f" % (x, t[x]["hbm_bytes"] / 1e9) for x in ("k_syrk", "k_build_AS", "k_fwdsub_lds", "k_chol_left", "k_search_mfma"))), s)

    def fr(key, which=ad):
        return which[key]["frame_us_median"]

    def cpu(key):
        return ad[key]["cpu_reference_us_median"]

    a = s.index("| scene | r05 | r06 | one CPU thread |")
    e = s.index("\n\n", a)
    s = s[:a] + ("| scene | r05 | r06 | one CPU thread |\n|---|---|---|---|\n"
                 "| shipped cfg, four known features | 99 us | **%.0f us** | %.0f us |\n"
                 "| a dozen features, mapping on | 170 us | **%.0f us** | %.0f us |\n"
                 "| 100 known features (configs[1]) | 214 us | **%.0f us** | %.0f us |" % (
                     fr("shipped_cfg_4_features"), cpu("shipped_cfg_4_features"), fr("mapping_on_dozen_features"), cpu("mapping_on_dozen_features"),
                     fr("configs1_100_features"), cpu("configs1_100_features"))) + s[e:]

    def sl_row(key):
        q = sl[key]
        return (q["ten_launches_direct"]["step_waited_us_median"], q["ten_launches_direct"]["step_queued_us"],
                q["fused_direct"]["step_waited_us_median"], q["fused_direct"]["step_queued_us"])
    rows = [("1 (4 features)", "batch1_4features_select4_capacity128"), ("1", "batch1_12features_select10_capacity128"),
            ("128", "batch128_12features_select10_capacity128"), ("256", "batch256_12features_select10_capacity128"),
            ("1024", "batch1024_12features_select10_capacity128")]
    tab = "| sequences | ten launches, waited / queued | fused, waited / queued |\n|---|---|---|\n" + "\n".join(
        "| %s | %.1f / %.1f | **%.1f / %.1f** |" % ((n,) + sl_row(kk)) for n, kk in rows)
    a = s.index("| sequences | ten launches, waited / queued | fused, waited / queued |")
    e = s.index("\n\n", a)
    s = s[:a] + tab + s[e:]

    s = re.sub(r"the torch path on one rank equals the plain run \([^)]*\)",
               "the torch path on one rank equals the plain run (%d k against %d k frames/s)" % (round(rc["value"] / 1e3), round(pl["value"] / 1e3)), s)
    s = re.sub(r"two ranks sharing the\ndevice under both launch forms \([^)]*\)",
               "two ranks sharing the\ndevice under both launch forms (%d k / %d k frames/s for 2 x 512 sequences, `parity.ranks_checked = 2`)" % (
                   round(tw["value"] / 1e3), round(tt["value"] / 1e3)), s)

    mlat = os.path.join(P, "r06_mapping_latency.json")
    if os.path.exists(mlat):
        ml = json.load(open(mlat))["plain"]
        mm = ad["mapping_on_dozen_features"]
        s = re.sub(r"\(in the committed run: [0-9.]+ / [0-9.]+ us; engine alone, `scripts/mapping_latency.py`: [0-9.]+ / [0-9.]+ us per waited step\)",
                   "(in the committed run: %.0f / %.0f us; engine alone, `scripts/mapping_latency.py`: %.0f / %.0f us per waited step)" % (
                       mm["frame_us_median_without_partial_feature"], mm["frame_us_median_with_partial_feature"],
                       ml["step_us_median_without_partial"], ml["step_us_median_with_partial"]), s)

    a = s.index("## 8. Measured on MI355X, round 6")
    e = s.index("## 9. Out of scope")
    hf = b.get("host_fed") or {}
    sec8 = ("## 8. Measured on MI355X, round 6 (library %s; `profiles/r06_final_*`)\n\n| what | value | file |\n|---|---|---|\n" % git +
            "| configs[2], driver protocol | %d k frames/s, %.3f ms per step | `r06_final_bench_driver.json` |\n" % (round(d["value"] / 1e3), d["ms_per_step"]) +
            "| configs[2], bench defaults (30 + 100 steps) | %d k frames/s, %.3f ms | `r06_final_bench.json` |\n" % (round(b["value"] / 1e3), b["ms_per_step"]) +
            "| parity in the run | every stepped frame of 8 sequences: RMSE %.1e m, final covariance %.1e relative | same |\n" % (
                b["parity"]["traj_rmse_vs_oracle"], b["parity"]["full_length"]["final_covariance_rel_fro"]) +
            "| CPU baseline (`port`: the oracle on the %d CPUs the cgroup grants, of %d hardware threads) | %d frames/s, single thread %d, scaling %.2f | same |\n" % (
                b["cpu_baseline"]["cores"], b["cpu_baseline"]["host_hardware_threads"], round(b["cpu_baseline"]["value"]),
                round(b["cpu_baseline"]["single_thread_frames_per_s"]), b["cpu_baseline"]["scaling"]) +
            "| host-fed (`host_fed` block) | %d k frames/s, overlap efficiency %.3f, link %.1f GB/s | same |\n" % (
                round(hf.get("frames_per_s", 0) / 1e3), hf.get("overlap_efficiency", 0), hf.get("h2d_GBps", 0)) +
            "| configs[1] (one sequence, 100 features) | %.0f us per step (%.0f without the event brackets) | `r06_final_bench_c1*.json` |\n" % (
                c1["ms_per_step"] * 1e3, c1n["ms_per_step"] * 1e3) +
            "| configs[3] per GPU (640x480, 200 features, 1024 sequences) | %.2f ms per step (`k_syrk` %.2f of the FP64 MFMA peak) | `r06_final_bench_c4.json`, `_c4_kernel_stats.csv` |\n" % (
                c4["ms_per_step"], c4["roofline"]["frac"]) +
            "| configs[4] per GPU (1280x720, 500 features, 512 sequences) | %.1f ms per step (`k_syrk` %.2f) | `r06_final_bench_c5.json`, `_c5_kernel_stats.csv` |\n" % (
                c5["ms_per_step"], c5["roofline"]["frac"]) +
            "| the reference's default workload (`--mapping`, 1024 sequences) | %.2f M frames/s, %.3f ms per step; maps equal on 16 sequences x 134 frames | `r06_final_bench_mapping.json` |\n" % (
                mp["value"] / 1e6, mp["ms_per_step"]) +
            "| a frame behind the adapter (four features / a dozen + mapping / 100) | %.0f / %.0f / %.0f us; with every frame uploaded instead of read in place: %.0f / %.0f / %.0f | `r06_adapter_latency.json`, `r06_adapter_latency_copy_frames.json` |\n" % (
                fr("shipped_cfg_4_features"), fr("mapping_on_dozen_features"), fr("configs1_100_features"),
                fr("shipped_cfg_4_features", adc), fr("mapping_on_dozen_features", adc), fr("configs1_100_features", adc)) +
            "| GPU tests / smoke | %s passed; 5.4e-15 / 2.0e-14 | `r06_final_pytest_gpu.log` |\n\n" % ntests)
    late = os.path.join(P, "r06_final_late_git.txt")
    if os.path.exists(late) and open(late).read().strip() != git:
        sec8 += ("The mapping line and the adapter, small-map and mapping latency files were taken at library %s; the others at %s,\n"
                 "which differs from it by the placement of the large matrices (section 3) - engines of those sizes are not placed.\n\n" % (
                     open(late).read().strip(), git))
    s = s[:a] + sec8 + s[e:]
    open(path, "w").write(s)
    print("DESIGN.md refreshed from profiles/ (library %s, %d bytes)" % (git, len(s)))


if __name__ == "__main__":
    main()
