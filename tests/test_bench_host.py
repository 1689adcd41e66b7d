"""CPU: the host-side arithmetic of bench.py (no GPU, no timing): the roofline object built from an event profile, the
algorithmic FLOP model behind `value -> TFLOP/s`, and the rule that `roofline.traffic` is only ever quoted from a PMC file
collected on exactly this build of the kernels."""
import json
import os

import pytest

import bench
from defensegan_amd import archs


def test_flop_model_matches_the_survey_numbers():
    """SURVEY appendix C / DESIGN 4: valid-tap MACs per latent row -- MNIST 524 288 + 7 372 800 + 8 388 608 + 67^2 * 64."""
    a = archs.make_arch("mnist")
    macs = archs.fwd_macs_per_row(a)
    assert macs == 128 * 4096 + 7372800 + 8388608 + 67 * 67 * 64
    # L forwards, L - 1 useful backwards, R restarts per image
    assert archs.flop_per_image(a, 10, 200) == 10 * 399 * 2.0 * macs
    assert archs.flop_per_image(a, 1, 1) == 2.0 * macs
    c = archs.make_arch("celeba")                           # 524 288 + 9 469 952 + 11 214 848 + 24 285 184 + 4 732 608
    assert archs.fwd_macs_per_row(c) == 524288 + 9469952 + 11214848 + 24285184 + 4732608 == 50226880
    # the per-image figures SURVEY quotes to six digits
    assert abs(archs.flop_per_image(c, 10, 200) - 4.00811e11) < 1e6
    assert abs(archs.flop_per_image(a, 10, 200) - 1.32252e11) < 1e6


def test_roofline_object_from_a_profile():
    prof = [
        {"name": "F2@gemm_batched_kernel<0, 3, 1>", "launches": 200, "ms": 200 * 0.275, "flops": 200 * 3.77e10},
        {"name": "B2@gemm_batched_kernel<0, 3, 1>", "launches": 199, "ms": 199 * 0.280, "flops": 199 * 3.77e10},
        {"name": "F3@gemm_batched_kernel<1, 2, 1>", "launches": 200, "ms": 200 * 0.311, "flops": 200 * 4.29e10},
        {"name": "UPD@momentum_update_kernel", "launches": 199, "ms": 199 * 0.007, "flops": 0.0},
        {"name": "never@launched", "launches": 0, "ms": 0.0, "flops": 0.0},
    ]
    kernels, r = bench.roofline_from_profile(prof, "not-a-profiled-workload", 256, 10, path_tflops=128.0)
    assert [k["name"] for k in kernels] == ["F2", "B2", "F3", "UPD"]
    assert r["kernel"] == "gemm_batched_kernel<0, 3, 1>"                 # most total time, grouped by symbol as rocprofv3 does
    avg_us = (200 * 275.0 + 199 * 280.0) / 399
    assert abs(r["avg_launch_us"] - avg_us) < 0.01
    assert abs(r["achieved"] - 3.77e10 / (avg_us * 1e-6) / 1e12) < 0.02
    assert abs(r["frac"] - r["achieved"] / 157.3) < 1e-3 and r["peak"] == 157.3 and r["bound"] == "mfma"
    assert r["traffic"] is None and r["traffic_source"] is None          # nothing collected for that workload
    assert abs(r["path_frac"] - 128.0 / 157.3) < 1e-3
    assert abs(r["sum_kernel_ms_per_step"] - sum(p["ms"] for p in prof)) < 1e-3
    # no profile (e.g. --no-profile, --strong): the whole-path rate stands in
    kernels, r = bench.roofline_from_profile([], "mnist", 256, 10, path_tflops=100.0)
    assert kernels == [] and r["kernel"] is None and r["achieved"] == 100.0 and r["traffic"] is None


def test_traffic_is_quoted_only_for_the_build_and_job_lists_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic: bytes of the LAYERS that ran as the dominant symbol (mean over them), from the committed PMC file --
    only for the same kernel sources (build id), the same row count and the same job lists (tuning id)."""
    rows = {"B3": {"kernel": "gemm_batched_kernel<0, 3, 1>", "bytes_per_launch": 100},
            "B2": {"kernel": "gemm_batched_kernel<0, 3, 1>", "bytes_per_launch": 300},
            "F3": {"kernel": "gemm_batched_kernel<1, 2, 1>", "bytes_per_launch": 1000}}
    doc = {"builds": {"mnist": bench.build_id(), "celeba": bench.build_id()}, "tuning_ids": {"mnist": "aaaaaaaaaaaa", "celeba": "bbbbbbbbbbbb"},
           "mnist": {"by_layer": rows}, "celeba": {"by_layer": {"B5": {"bytes_per_launch": 7}}}}
    prof_dir = tmp_path / "profiles"
    prof_dir.mkdir()
    (prof_dir / bench.TRAFFIC_FILE).write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.traffic_for("mnist", ["B3", "B2"], 256, 10, "aaaaaaaaaaaa") == (200, "profiles/" + bench.TRAFFIC_FILE)
    assert bench.traffic_for("mnist", ["F3"], 256, 10, "aaaaaaaaaaaa")[0] == 1000
    assert bench.traffic_for("fmnist", ["B3"], 256, 10, "aaaaaaaaaaaa")[0] == 100      # same architecture and row count
    assert bench.traffic_for("celeba", ["B5"], 128, 10, "bbbbbbbbbbbb")[0] == 7
    assert bench.traffic_for("mnist", ["B3"], 256, 10, "cccccccccccc") == (None, None)    # other job lists ran than were profiled
    assert bench.traffic_for("mnist", ["B3"], 50, 10, "aaaaaaaaaaaa") == (None, None)     # another row count: not measured
    assert bench.traffic_for("mnist", ["B3", "B9"], 256, 10, "aaaaaaaaaaaa") == (None, None)    # a layer the file does not hold
    assert bench.traffic_for("mnist", [], 256, 10, "aaaaaaaaaaaa") == (None, None)
    assert bench.traffic_for("mnist_bn", ["B3"], 256, 10, "aaaaaaaaaaaa") == (None, None)  # --use_bn runs quote nothing
    doc["builds"]["celeba"] = "0" * 12                                  # CelebA collected on other kernel sources
    (prof_dir / bench.TRAFFIC_FILE).write_text(json.dumps(doc))
    assert bench.traffic_for("mnist", ["B3"], 256, 10, "aaaaaaaaaaaa")[0] == 100
    assert bench.traffic_for("celeba", ["B5"], 128, 10, "bbbbbbbbbbbb") == (None, None)


def test_build_id_covers_every_kernel_source_and_header():
    """The id must change whenever anything compiled into the library changes: every file under csrc/ is hashed."""
    from defensegan_amd import build
    listed = {os.path.basename(f) for f in build.SOURCES + build.HEADERS}
    on_disk = {f for f in os.listdir(build.CSRC) if f.endswith((".hip", ".cpp", ".h"))}
    assert on_disk <= listed, sorted(on_disk - listed)
    assert len(bench.build_id()) == 12 and int(bench.build_id(), 16) >= 0


def test_build_id_ignores_comments_and_layout_only():
    """The id is taken over the sources without comments (a corrected comment must not orphan the measurements made on
    that code) -- and over nothing less: the stripped text equals what gcc's own comment removal leaves."""
    import shutil
    import subprocess
    from defensegan_amd import build
    s = build._strip_comments
    assert s('int a = 1; // c\n/* b\n b */ const char* u = "http://x/*y*/"; char q = \'"\'; // t\n') == \
        'int a = 1; const char* u = "http://x/*y*/"; char q = \'"\';'
    assert s("a/**/b") == "a b" and s("x = 1;   \n\n  y = 2;") == "x = 1; y = 2;"
    assert s('p = "a \\" // not a comment"; // yes') == 'p = "a \\" // not a comment";'
    assert s("int a = 1;") != s("int a = 2;")
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc to compare with")
    for f in build.SOURCES + build.HEADERS:
        path = os.path.join(build.CSRC, f)
        ref = subprocess.run([gcc, "-fpreprocessed", "-dD", "-E", "-P", "-x", "c++", path], capture_output=True, text=True,
                             check=True).stdout
        with open(path, encoding="utf-8") as fh:
            mine = s(fh.read())
        assert mine.replace("#pragma once ", "", 1) == " ".join(ref.split()), f      # gcc consumes the pragma


def test_committed_traffic_file_is_well_formed():
    path = os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "profiles", bench.TRAFFIC_FILE)
    if not os.path.exists(path):
        pytest.skip("no PMC traffic file committed")
    with open(path) as fh:
        doc = json.load(fh)
    for wl in ("mnist", "celeba"):
        assert len(doc["builds"][wl]) == 12 and len(doc["tuning_ids"][wl]) == 12
        rows = doc[wl]["by_layer"]
        assert {"F2", "F3", "B3", "B2"} <= set(rows), wl
        for layer, rec in rows.items():
            assert rec["bytes_per_launch"] > 0 and rec["launches_profiled"] > 0 and rec["kernel"], (wl, layer)
            # FETCH_SIZE (KB, doubled per the gfx950 note) + WRITE_SIZE (KB), per launch
            assert abs(rec["bytes_per_launch"] - (2.0 * rec["fetch_kb_raw"] + rec["write_kb"]) * 1024) <= 2048, (wl, layer)


def test_pmc_rows_are_labelled_by_the_launch_order_of_an_iteration():
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "tools", "pmc_traffic.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    g, lin = "gemm_batched_kernel<0, 3, 1>", "lin_stationary_kernel<4, 2>"
    it = [lin, g, g, "mnist_tail_pipe_kernel<64>", g, g, "lin_stationary_kernel<8, 0>", "momentum_update_kernel"]
    last = [lin, g, g, "mnist_tail_mfma_kernel<64>", "select_kernel"]
    assert m.layer_labels(it + it + last, "mnist") == ["F1", "F2", "F3", "T5fb", "B3", "B2", "B1", "UPD"] * 2 + ["F1", "F2", "F3", "T5", ""]
    c = [g] * 4 + ["celeba_tail_fwd_split_kernel<64>", "celeba_tail_bwd_persist_kernel<64>"] + [g] * 4 + ["momentum_update_kernel"]
    assert m.layer_labels(c, "celeba") == ["F1", "F2", "F3", "F5", "T6f", "T6b", "B5", "B3", "B2", "B1", "UPD"]

def test_gpus_n_never_degrades_to_a_one_gpu_run():
    """`--gpus N` either runs N ranks or exits: launcher present -> WORLD_SIZE must equal N; launcher absent and N > 1 ->
    the bench re-executes itself under torch.distributed.run; fewer than N devices -> SystemExit, whatever the environment."""
    r = bench.resolve_ranks
    assert r(1, {}, 1) == (1, 0, 0, False)
    assert r(8, {}, 8) == (8, 0, 0, True)                                   # no launcher: relaunch with 8 ranks
    assert r(2, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, 8) == (2, 1, 1, False)
    assert r(1, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, 1) == (1, 0, 0, False)   # DG_BENCH_FORCE_DIST runs
    # a launcher that narrows each rank's visibility to its own GPU (one device visible, LOCAL_RANK beyond it): device 0
    assert r(8, {"WORLD_SIZE": "8", "RANK": "5", "LOCAL_RANK": "5"}, 1) == (8, 5, 0, False)
    for n, env, ndev in [(8, {}, 1), (2, {}, 0), (8, {"WORLD_SIZE": "1"}, 8), (2, {"WORLD_SIZE": "4"}, 8),
                         (2, {"WORLD_SIZE": "2", "LOCAL_RANK": "1"}, 0), (0, {}, 1),
                         (2, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "5"}, 4)]:
        with pytest.raises(SystemExit) as e:
            r(n, env, ndev)
        assert "bench.py" in str(e.value)                                   # a message, i.e. a non-zero exit status


def test_self_launch_command_is_the_drivers_launch_line():
    cmd = bench.self_launch_command(["--gpus", "4", "--steps", "3", "--warmup", "1"], 4, 29544)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29544"
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_gpus_2_on_a_box_with_fewer_gpus_exits_nonzero_with_a_message():
    """The real command line, as a process: on this CPU-only container (and on a 1-GPU box) `--gpus 2` must fail loudly
    instead of printing a JSON line."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, bench.__file__, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 2" in p.stderr and "GPU(s) are visible" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_self_launch_starts_one_rank_per_gpu(tmp_path):
    """The relaunch line really starts N workers with RANK / LOCAL_RANK / WORLD_SIZE set: run it on a stand-in script
    (torch.distributed.run, 2 processes, CPU) and read what each rank saw."""
    import subprocess
    import sys
    script = tmp_path / "worker.py"
    script.write_text("import os, sys\n"
                      "open(os.path.join(sys.argv[1], 'rank%s' % os.environ['RANK']), 'w').write("
                      "'%s %s %s %s' % (os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'], os.environ['MASTER_ADDR'], ' '.join(sys.argv[2:])))\n")
    cmd = bench.self_launch_command([str(tmp_path), "--gpus", "2"], 2, bench._free_port())
    cmd[cmd.index(os.path.abspath(bench.__file__))] = str(script)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert (tmp_path / "rank0").read_text() == "2 0 127.0.0.1 --gpus 2"
    assert (tmp_path / "rank1").read_text() == "2 1 127.0.0.1 --gpus 2"


def test_kernel_trace_rows_are_labelled_with_the_bench_lines_layers(tmp_path):
    """tools/kernel_trace_by_layer.py: launches grouped by (symbol, grid, LDS) and labelled with the layer of bench.py's
    per-layer rows -- one symbol serving two layers is told apart by duration, the Linear kernels are matched although the trace
    spells their defaulted template argument (`lin_stationary_kernel<4, 2, false>`), and the short update launch although the
    hipEvent markers add several microseconds to its bench row."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.abspath(bench.__file__))
    rows = [("void dg::gemm_batched_kernel<0, 3, 1>(dg::GemmArgs)", 2384 * 256, 49152, 311000),
            ("void dg::gemm_batched_kernel<0, 3, 1>(dg::GemmArgs)", 1472 * 256, 49152, 279000),
            ("void dg::(anonymous namespace)::lin_stationary_kernel<4, 2, false>(dg::LinArgs)", 512 * 256, 49152, 28000),
            ("void dg::momentum_update_kernel(float*, float*, float const*, int, long long, int, float, float, float*)", 320 * 256, 0, 6100),
            ("void dg::gemm_batched_kernel<0, 3, 1>(dg::GemmArgs)", 999 * 256, 49152, 300000)]          # a candidate list: 3 launches only
    trace = tmp_path / "kernel_trace.csv"
    with open(trace, "w") as fh:
        fh.write("Kernel_Name,Grid_Size_X,Workgroup_Size_X,LDS_Block_Size,Start_Timestamp,End_Timestamp\n")
        t = 0
        for rep in range(70):
            for i, (name, grid, lds, ns) in enumerate(rows):
                if i == 4 and rep >= 3:
                    continue
                fh.write('"%s",%d,256,%d,%d,%d\n' % (name, grid, lds, t, t + ns + rep % 3))
                t += ns + 5000
    line = {"kernels": [{"name": "B3", "kernel": "gemm_batched_kernel<0, 3, 1>", "avg_us": 314.0},
                        {"name": "B2", "kernel": "gemm_batched_kernel<0, 3, 1>", "avg_us": 282.0},
                        {"name": "F1", "kernel": "lin_stationary_kernel<4, 2>", "avg_us": 33.5},
                        {"name": "UPD", "kernel": "momentum_update_kernel", "avg_us": 11.5}]}
    bj = tmp_path / "bench.json"
    bj.write_text(json.dumps(line) + "\n")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_trace_by_layer.py"), str(trace), str(bj)],
                         capture_output=True, text=True, check=True).stdout.splitlines()
    got = {}
    import csv
    for r in csv.DictReader(out):
        got[(r["Layer"], r["Kernel"])] = (int(r["Calls"]), float(r["AverageNs"]))
    assert got[("B3", "gemm_batched_kernel<0, 3, 1>")][0] == 70 and abs(got[("B3", "gemm_batched_kernel<0, 3, 1>")][1] - 311001) < 2
    assert got[("B2", "gemm_batched_kernel<0, 3, 1>")][0] == 70
    assert got[("F1", "lin_stationary_kernel<4, 2, false>")][0] == 70
    assert got[("UPD", "momentum_update_kernel")][0] == 70
    assert got[("", "gemm_batched_kernel<0, 3, 1> [candidate job lists, not kept]")][0] == 3


def test_rank_cpus_are_disjoint_contiguous_slices():
    cpus = set(range(64))
    got = [bench.rank_cpus(r, 8, cpus) for r in range(8)]
    assert got[0] == list(range(0, 8)) and got[3] == list(range(24, 32)) and got[7] == list(range(56, 64))
    assert len({c for g in got for c in g}) == 64
    assert bench.rank_cpus(1, 2, set(range(256))) == list(range(128, 136))          # at most 8 CPUs per rank
    assert bench.rank_cpus(5, 8, {0, 1, 2}) == [0, 1, 2]                            # fewer CPUs than ranks: no empty set
    phys, n = bench.physical_cores_one_socket()
    assert n >= 1 and (phys is None or 1 <= phys <= n)


def test_dominant_symbol_is_the_one_with_the_most_flop_ties_go_to_the_forward_layer():
    """Generator.3's forward and backward run as two symbols with the same FLOP per launch and durations 0.05 % apart: the
    roofline object must not depend on which of the two happened to take longer in the profiled step."""
    def prof(f3_ms, b3_ms, n_b=199):
        return [{"name": "F2@gemm<0, 2, 1>", "launches": 200, "ms": 54.0, "flops": 200 * 3.77e10},
                {"name": "F3@gemm<1, 2, 1>", "launches": 200, "ms": f3_ms, "flops": 200 * 4.29e10},
                {"name": "T5fb@tail", "launches": 199, "ms": 10.2, "flops": 199 * 2.9e9},
                {"name": "B3@gemm<0, 3, 1>", "launches": n_b, "ms": b3_ms, "flops": n_b * 4.29e10},
                {"name": "UPD@upd", "launches": 199, "ms": 1.2, "flops": 0.0}]
    for f3, b3 in ((62.06, 62.03), (62.00, 62.40)):
        _, r = bench.roofline_from_profile(prof(f3, b3), "mnist", 256, 10, 133.0, None)
        assert r["kernel"] == "gemm<1, 2, 1>" and r["layers"] == ["F3"]
        assert [m["layer"] for m in r["mfma_layers"][:3]] in (["F3", "B3", "F2"], ["B3", "F3", "F2"])
        assert abs(r["achieved"] - 200 * 4.29e10 / (f3 * 1e-3) / 1e12) < 0.01
    _, r = bench.roofline_from_profile(prof(62.0, 62.0, n_b=200), "mnist", 256, 10, 133.0, None)      # exact tie -> forward
    assert r["layers"] == ["F3"]


def test_cpu_baseline_follows_the_fixed_rule():
    """`cpu_baseline`: a fixed thread count (CPU_BASELINE_THREADS, capped by the box), two timed batches, the faster one and the
    spread -- run here at a toy size (4 images, R = 2, L = 2) so that the CPU suite stays short."""
    import numpy as np
    from defensegan_amd import synth
    p = synth.make_weights("mnist", seed=1234, gain=2.0, bias_range=0.0)
    x = np.random.RandomState(0).uniform(0, 1, size=(4, 28, 28, 1)).astype(np.float32)
    old = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    r = bench.cpu_baseline("mnist", p, x, R=2, L=2)
    assert r["kind"] == "port" and r["unit"] == "images/s" and r["value"] > 0
    assert r["threads"] == r["cores"] == min(bench.CPU_BASELINE_THREADS, os.cpu_count(), bench.physical_cores_one_socket()[0] or os.cpu_count())
    assert 0.0 <= r["spread"] and "fixed rule" in r["sample"] and "two batches" in r["sample"]
    if old is not None:
        assert set(os.sched_getaffinity(0)) == old                      # the pinning is undone
