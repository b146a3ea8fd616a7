"""Host-side checks of the measurement plumbing: the kernel -> family rules behind
``roofline.traffic``, the recomputation of the committed traffic files from the committed counter sums, and the way
``bench.py`` chooses the roof a kernel family is priced against."""
import csv
import glob
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


pmc = _load(os.path.join(ROOT, "tools", "pmc_traffic.py"), "pmc_traffic")
bench = _load(os.path.join(ROOT, "bench.py"), "bench_module")


@pytest.mark.parametrize("stats", sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[34]_bench_*_kernel_stats_*.csv"))),
                         ids=os.path.basename)
def test_every_profiled_kernel_maps_to_a_family_with_launches(stats):
    """Round 3 dropped 1.55 GB per step: ``k_wgrad_reduce4`` matched no family rule, became a family of its own and was
    given 0 launches.  Every kernel name of the committed rocprofv3 ``--stats`` files must land in a family that has a
    launch-counting kernel in the same file, or be a declared helper of one."""
    bf16s = "bf16s" in os.path.basename(stats)
    launches, members = {}, {}
    with open(stats) as f:
        for r in csv.DictReader(f):
            fam = pmc.family(r["Name"], bf16s)
            members.setdefault(fam, []).append(r["Name"])
            launches[fam] = launches.get(fam, 0) + (int(r["Calls"]) if pmc.counts_as_launch(r["Name"]) else 0)
    assert launches, stats
    helper_only = set()
    for fam, n in launches.items():
        if n == 0 and all(not pmc.counts_as_launch(m) for m in members[fam]):
            helper_only.add(fam)    # untagged helpers in a --stats file (no dispatch order there): see below
            continue
        assert n > 0, (fam, members[fam])
    # a --stats csv has no dispatch order, so its reduce rows cannot be tagged with the kernel they followed: they fall to the
    # default family - which may have no launch of its own in a workload that only runs the other weight-gradient kernels
    # (deep-12); what must hold is that SOME weight-gradient kernel ran
    if helper_only:
        sfx = ", bf16s" if bf16s else ""
        assert helper_only <= {f + sfx for f in pmc.WGRAD_FAMILIES}
        assert any(launches.get(f + sfx, 0) > 0 for f in pmc.WGRAD_FAMILIES), stats
    for fam, names in members.items():
        for name in names:
            if not pmc.counts_as_launch(name):   # a helper: declared, and its family has a main kernel in this file
                assert any(pmc.base_name(name).startswith(h) and pmc.HELPERS[h] == fam.split(",")[0] for h in pmc.HELPERS)
    reduces = [n for names in members.values() for n in names if "k_wgrad_reduce" in n]
    for name in reduces:
        assert pmc.family(name, bf16s).startswith("k_conv_wgrad") and not pmc.counts_as_launch(name)


def test_event_frame_weight_gradient_instances_belong_to_the_weight_gradient_family():
    """``k_conv_first<CIN, KS, WGRAD, BNAPPLY, SB>``: the third template argument decides (round 3's rule looked for a
    trailing ``true>`` and filed the weight gradient ``<2, 3, true, true, false>`` under the forward kernel: 1.6 GB per step in
    the wrong row)."""
    fwd = "void (anonymous namespace)::k_conv_first<2, 3, false, false, false>(float const*, float const*)"
    wg = "void (anonymous namespace)::k_conv_first<2, 3, true, true, false>(float const*, float const*)"
    wg_sb = "void (anonymous namespace)::k_conv_first<2, 3, true, false, true>(float const*, float const*)"
    assert pmc.family(fwd, False) == "k_conv_first<2, 3, false>"
    assert pmc.family(wg, False) == "k_conv_first<wgrad>" and pmc.family(wg_sb, True) == "k_conv_first<wgrad>, bf16s"
    assert pmc.counts_as_launch(wg)


def test_reduce_launches_are_attributed_to_the_weight_gradient_kernel_they_follow(tmp_path):
    """The raw pass tags every split-K reduce with the weight-gradient kernel it followed in dispatch order: the three
    kernels are priced separately and each carries its own reduce bytes."""
    raw = tmp_path / "f_counter_collection.csv"
    rows = [(3, "void k_wgrad_reduce4(float const*)", 10.0), (1, "void k_conv_wgrad_halo<2, 4, 1, 3, false>(float const*)", 100.0),
            (2, "void k_wgrad_reduce4(float const*)", 7.0), (4, "void k_conv_wgrad_pipe<2, 2, 2, 2, 32, false, false, false>(x)", 50.0),
            (5, "void k_wgrad_reduce4(float const*)", 3.0), (6, "void k_wgrad_reduce4(float const*)", 2.0),
            (7, "void k_conv_first<2, 3, true, true, false>(x)", 20.0), (8, "void k_wgrad_reduce<1>(x)", 1.0),
            (9, "void k_affine_neuron_fwd<1, 4, 1, false>(x)", 5.0)]
    with open(raw, "w") as f:
        f.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n")
        for d, n, v in rows:
            f.write(f'{d},"{n}",FETCH_SIZE,{v}\n')
    per = pmc.aggregate(str(raw), "FETCH_SIZE")
    fam = pmc.by_family(per, False)
    assert fam["k_conv_wgrad_halo"] == [117.0, 1]          # dispatches 1, 2, 3: the kernel + both reduce passes
    assert fam["k_conv_wgrad_pipe"] == [55.0, 1]
    assert fam["k_conv_first<wgrad>"] == [21.0, 1]
    assert fam["k_affine_neuron_fwd<1>"] == [5.0, 1]
    counters = tmp_path / "c.csv"
    pmc.write_counters(str(counters), per, {})
    again, _ = pmc.read_counters(str(counters))
    assert pmc.by_family(again, False) == fam                # the committed counter sums keep the attribution


def test_a_helper_without_its_main_kernel_is_an_error_not_a_silent_drop():
    fetch = {"void k_wgrad_reduce4<4>(float*)": [100.0, 3]}
    with pytest.raises(RuntimeError, match="no launch-counting kernel"):
        pmc.traffic_table(fetch, {}, steps=1, bf16s=False)
    fetch["void k_conv_wgrad_pipe<64, 64>(float*)"] = [1000.0, 2]
    rows, meta = pmc.traffic_table(fetch, {"void k_conv_wgrad_pipe<64, 64>(float*)": [10.0, 2]}, steps=1, bf16s=False)
    fam = rows["k_conv_wgrad_pipe"]
    assert fam["launches"] == 2                                        # the reduce adds bytes, not launches
    assert fam["read_bytes_per_launch"] == 2 * 1024 * 1100.0 / 2       # FETCH_SIZE: KiB, doubled on gfx950
    assert fam["write_bytes_per_launch"] == 1024 * 10.0 / 2
    assert meta["hbm_bytes_per_step"] == 2 * 1024 * 1100.0 + 1024 * 10.0


@pytest.mark.parametrize("traffic", sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_pmc_traffic_*.json"))),
                         ids=os.path.basename)
def test_committed_traffic_files_recompute_from_the_committed_counter_sums(traffic):
    """``roofline.traffic`` = (tile + reduce bytes) / launches, recomputed here from ``profiles/r04_pmc_counters_*.csv``."""
    data = json.load(open(traffic))
    meta = data["_meta"]
    counters = os.path.join(ROOT, "profiles", meta["counters"])
    assert os.path.exists(counters), counters
    fetch, write = pmc.read_counters(counters)
    rows, remeta = pmc.traffic_table(fetch, write, meta["profiled_steps"], bf16s=meta.get("bf16s", False))
    assert set(rows) == {k for k in data if k != "_meta"}
    for k, v in rows.items():
        assert v["launches"] == data[k]["launches"], k
        assert v["hbm_bytes_per_launch"] == pytest.approx(data[k]["hbm_bytes_per_launch"], rel=1e-12), k
    assert remeta["hbm_bytes_per_step"] == pytest.approx(meta["hbm_bytes_per_step"], rel=1e-12)
    # the weight-gradient family carries its reduce kernels' bytes
    wg = [n for n in fetch if "k_wgrad_reduce" in n]
    if wg:
        sfx = ", bf16s" if meta.get("bf16s") else ""
        fams = [f + sfx for f in pmc.WGRAD_FAMILIES if f + sfx in data]
        tiles = [n for n in fetch if pmc.family(n, meta.get("bf16s", False)) in fams and pmc.counts_as_launch(n)]
        only_tiles = sum(2 * 1024 * fetch[n][0] + 1024 * write.get(n, [0.0])[0] for n in tiles)
        assert sum(data[f]["hbm_bytes_per_launch"] * data[f]["launches"] for f in fams) > only_tiles


def test_bound_is_chosen_by_arithmetic_intensity_not_by_kernel_name():
    # the weight-gradient family of the GEN1 step: 26.3 GFLOP and 377 MB per launch = 70 FLOP/B, ridge of bf16 x 3 = 104
    row = {"flops": 43 * 26.32e9, "bytes": 43 * 377e6, "tflops": 180.0, "gbs": 2570.0}
    head = bench.roofline_head("k_conv_wgrad_pipe", row, "fp16x3", "bf16x3", sb=False)
    assert head["bound"] == "hbm" and head["unit"] == "GB/s" and head["frac"] == pytest.approx(2570.0 / 8000.0)
    assert head["frac_mfma"] == pytest.approx(180.0 / (2500.0 / 3)) and 69 < head["intensity_flop_per_byte"] < 71
    assert 104 < head["ridge_flop_per_byte"] < 105
    # a 128 -> 128 3x3 layer: 2*9*128*128 / (2*128*4) = 288 FLOP/B: matrix-bound against the same peak
    row = {"flops": 288.0e9, "bytes": 1.0e9, "tflops": 290.0, "gbs": 1000.0}
    head = bench.roofline_head("k_conv_halo3<128, fwd>", row, "fp16x3", "bf16x3", sb=False)
    assert head["bound"] == "mfma" and head["frac"] == pytest.approx(290.0 / (2500.0 / 3))
    # exact-fp32 arithmetic moves the ridge to 157.3 / 8 = 19.7 FLOP/B: the same weight gradients are matrix-bound there
    head = bench.roofline_head("k_conv_wgrad_pipe", {"flops": 70.0, "bytes": 1.0, "tflops": 100.0, "gbs": 1400.0}, "fp32", "fp32", False)
    assert head["bound"] == "mfma" and head["peak"] == pytest.approx(157.3)
    # scans are priced against HBM
    head = bench.roofline_head("k_affine_neuron_fwd<1>", {"flops": 1.0, "bytes": 1.0, "tflops": 4.0, "gbs": 5300.0}, "fp16x3", "bf16x3", False)
    assert head["bound"] == "hbm" and head["frac"] == pytest.approx(5300.0 / 8000.0)


def test_chain_row_prices_the_scans_at_the_ideal_fusion_bytes():
    n = 844.5e6   # neuron-timesteps of one GEN1 step (5 278 080 x 5 x 32)
    table = {"k_affine_neuron_fwd<1>": {"ms": 2 * 2.09, "bytes": 2 * 12.0 * n},
             "k_affine_neuron_bwd<1>": {"ms": 2 * 2.65, "bytes": 2 * 16.0 * n},
             "k_bn_bwd_apply": {"ms": 2 * 1.34, "bytes": 2 * 12.0 * n},
             "k_conv_wgrad_pipe": {"ms": 12.0, "bytes": 1.0}}
    row = bench.chain_row(table, 2 * n, 2, sb=False)
    assert row["ms_per_step"] == pytest.approx(6.08) and row["traffic_ratio"] == pytest.approx(2.0)
    assert row["moved_bytes_per_neuron_timestep"] == pytest.approx(40.0)
    assert row["frac_hbm"] == pytest.approx(20.0 * n / 6.08e-3 / 8e12, rel=1e-6) and 0.34 < row["frac_hbm"] < 0.36
    assert bench.chain_row(table, 2 * n, 2, sb=True)["ideal_bytes_per_neuron_timestep"] == 10.0
