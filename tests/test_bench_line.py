"""bench.py: the roofline arithmetic is reproducible from the files under profiles/ (CPU), and a small run prints one
JSON line with the fields the bench contract names (GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_valu_roofline_is_recomputed_from_profiles():
    import bench
    isa = json.load(open(os.path.join(ROOT, "profiles", "k_ens_block_isa.json")))["variants"]["512,20,3"]
    rates = json.load(open(os.path.join(ROOT, "profiles", "valu_issue_rate.json")))
    c = {"block_tpb": 512, "block_npt": 20, "block_enc_lds": 3}
    units_per_s = 1.4e12
    v = bench.valu_roofline(c, units_per_s)
    cls = isa["by_class"]
    cycles = (cls["packed"] * rates["v_pk_fma_f32"]["2"] + cls["trans"] * rates["v_rcp_f32"]["2"] +
              cls["plain_fma"] * rates["v_fma_f32"]["2"] + cls["plain"] * rates["v_max_i32"]["2"])
    peak = 1024 * 2.4e9 * 64 * 20 / cycles
    assert v["variant"] == "512,20,3" and v["waves_per_simd"] == 2
    assert abs(v["peak_neuron_steps_per_s"] - peak) / peak < 1e-3
    assert abs(v["frac"] - units_per_s / peak) < 2e-3
    assert v["valu_instructions_per_wave_timestep"] == sum(cls[k] for k in ("packed", "trans", "plain_fma", "plain"))
    assert abs(isa["valu_per_neuron_step"] * 20 - v["valu_instructions_per_wave_timestep"]) < 1e-6
    # a variant without a committed instruction count is reported as such, not priced with another variant's numbers
    assert "note" in bench.valu_roofline({"block_tpb": 256, "block_npt": 40, "block_enc_lds": 0}, units_per_s)


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--ssp-dim", "55", "--pi-n-neurons", "500", "--steps", "2",
                        "--warmup", "1", "--block", "256", "--cpu-steps", "20", "--cpu-warmup", "10", "--slam-steps", "0"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "sim-sec/wall-sec" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 2 * 256 * 0.001 / (2 * d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("valu", "hbm", "mfma") and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["parity"]["max_cosine_error"] < 1e-3 and d["value_end_to_end"] > 0
