"""The numbers DESIGN.md and README.md quote are the ones the committed profiles hold (no GPU): a figure in prose that no
record under profiles/ backs is how a stale claim survives a round."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def record(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


def text(name):
    return open(os.path.join(ROOT, name), encoding="utf-8").read()


def shown(x):
    """11666.72 -> '11,667'; 9277.9 -> '9278' (the documents write four digits without a comma)."""
    n = int(round(x))
    return f"{n:,}" if n >= 10000 else str(n)


def test_design_results_table_is_the_committed_records():
    d = text("DESIGN.md")
    c3, interp = record("r6_spec_c3_plain_bench.json"), record("r6_interp_c3_bench.json")
    c2, c4, orbit = (record(f"r6_spec_{w}_1gpu_bench.json") for w in ("c2", "c4", "orbit"))
    rows = {
        "C3": " / ".join(shown(c3[k]) for k in ("value", "value_new_view", "value_moving_camera_2_in_flight")),
        "interp": shown(interp["value"]),
        "C2": " / ".join(shown(c2[k]) for k in ("value", "value_new_view")),
        "C4": " / ".join(shown(c4[k]) for k in ("value", "value_new_view", "value_moving_camera_2_in_flight")),
        "orbit": shown(orbit["value"]),
        "host": " / ".join(shown(c3[k]) for k in ("value_through_render_thread_as_main_c_calls_it", "value_through_render_thread_pipelined")),
    }
    table = d[d.index("### 3.8 Results"):d.index("## 4. Multi-GPU")]
    for what, want in rows.items():
        assert want in table, (what, want)
    # one kernel behind every specialised C3 / C4 / orbit record, and it is the one the table names
    keys = {r["config"]["kernel_key"] for r in (c3, c4, orbit)}
    assert len(keys) == 1 and keys.pop() in table
    assert "%.1f" % c3["cpu_baseline"]["value"] in table


def test_roofline_paragraph_is_the_committed_record():
    d = text("DESIGN.md")
    c3 = record("r6_spec_c3_plain_bench.json")
    para = d[d.index("### 3.7 Roofline"):d.index("### 3.8 Results")]
    r, v = c3["roofline"], c3["valu"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["algorithmic_bytes"] == 4 * 3840 * 2160
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert "%.1f GB/s" % r["achieved"] in para and "%.2f %%" % (100 * r["frac"]) in para
    assert "%.2f cycles" % v["cycles_per_valu_instruction"] in para and "`issue_frac` %.2f" % v["issue_frac"] in para
    assert "%d VALU" % int(v["valu_instructions_per_pixel"] + 0.5) in para
    assert "%.2f× algorithmic" % (r["traffic"] / r["algorithmic_bytes"]) in para
    # the counters were taken on the code object the line is about
    pmc = record("pmc_traffic.json")["kernels"]["lol_render_spec"]
    assert pmc["kernel_key"] == c3["config"]["kernel_key"]
    # rocprofv3's kernel-trace average agrees with the HIP events of the same command (the contract's cross-check)
    stats = open(os.path.join(ROOT, "profiles", "r6_spec_c3_kernel_stats.csv")).read()
    avg_ns = float(re.search(r'^"lol_render_spec",\d+,\d+,([0-9.]+)', stats, re.M).group(1))
    under_trace = record("r6_spec_c3_bench.json")["roofline"]["kernel_ms_avg"]
    assert abs(avg_ns / 1e6 - under_trace) / under_trace < 0.03
    assert "%.3f ms over" % (avg_ns / 1e6) in para


def test_every_whole_frame_record_found_no_differing_pixel():
    names = ["r6_spec_c3_plain_bench.json", "r6_spec_c2_1gpu_bench.json", "r6_spec_c4_1gpu_bench.json"]
    names += [f"r6_spec_{w}_fixed_order_bench.json" for w in ("c2", "c3", "c4")]
    for n in names:
        r = record(n)
        p = r["cpu_baseline"]["parity_vs_gpu"]
        assert p["pixels_compared"] == r["config"]["width"] * r["config"]["height"] and p["pixels_differing"] == 0, n
        assert r["config"]["env"]["lol_gpu_tuning_switches"] in (None, "", "LOL_GPU_CACHE_ANY_COMPILER=1"), n
    for n in (f"r6_spec_{w}_fixed_order_bench.json" for w in ("c2", "c3", "c4")):
        assert record(n)["tile_order"] in ("rows", "cols"), n
    orbit = record("r6_spec_orbit_1gpu_bench.json")
    found = json.dumps(orbit)
    assert '"pixels_differing": 0' in found and '"pixels_differing": 1' not in found


def test_readme_headline_is_the_committed_record():
    r = text("README.md")
    c3 = record("r6_spec_c3_plain_bench.json")
    assert "**%s Mpixels/s**" % shown(c3["value"]) in r
    assert "**%s for a single frame" % shown(c3["value_new_view"]) in r
    assert "a host sees %s" % shown(c3["value_through_render_thread_as_main_c_calls_it"]) in r
