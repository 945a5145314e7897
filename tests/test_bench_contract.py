"""bench.py's host-side pieces on the CPU: the driver's command line parses to the declared C3 workload, the SURVEY 8(d)
byte model is the one the roofline quotes, the launcher line for --gpus N is the driver's, and the product never needs
the oracle before the timed region (the only import of `oracle` sits in cpu_baseline)."""
import ast
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv):
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        spec.loader.exec_module(mod)
        return mod, mod.parse()
    finally:
        sys.argv = old


def test_default_and_driver_command_lines_name_the_declared_c3():
    mod, a = _bench([])
    assert (a.gpus, a.steps, a.warmup) == (1, 5, 1)  # no flags: one GPU, a handful of steps
    assert a.runs == mod.DECLARED_RUNS == 1_000_000_000 and a.sigma == 253 and a.reads == 10_000_000
    assert (a.read_len, a.bp_per_read, a.out_bits, a.scaling) == (44, 200, 16, "weak")
    assert a.want("dna_m200") and a.want("c4_ms_doc") and a.want("long_reads_c5") and a.want("real_bwt_digest_walk")
    assert a.want("cli_file_to_file") and not a.want("real_bwt_large")  # (the large real BWT is opt-in: minutes of suffix sorting)
    assert 0 < a.cpu_seconds <= 5 and a.wall_budget > 240
    mod, a = _bench(["--gpus", "8", "--steps", "20", "--warmup", "5"])  # the driver's line
    assert (a.gpus, a.steps, a.warmup, a.runs) == (8, 20, 5, 1_000_000_000)
    mod, a = _bench(["--stand-in", "--no-extras", "--scaling", "strong"])
    assert a.runs == 1 << 28 and not a.want("dna_m200") and a.scaling == "strong"
    mod, a = _bench(["--legs", "positive_100,c4_ms_doc"])
    assert a.want("c4_ms_doc") and a.want("positive_100") and not a.want("dna_m200")
    mod, a = _bench(["--legs", "real_bwt_large"])
    assert a.want("real_bwt_large") and not a.want("cli_file_to_file")


def test_bytes_per_step_is_survey_8d():
    mod, a = _bench([])
    # 64 * (1 + 2 f_mis + f_pred) + 1 + out: the headline mix (f_mis 0.6138, f_pred 0.2875, u16 values) = 163.97
    b, f_mis, f_pred = mod.bytes_per_step({"steps": 10_000, "jumps": 6_138, "pred_jumps": 2_875}, 2)
    assert abs(b - 163.97) < 0.01 and abs(f_mis - 0.6138) < 1e-9 and abs(f_pred - 0.2875) < 1e-9
    assert mod.bytes_per_step({"steps": 100, "jumps": 0, "pred_jumps": 0}, 4)[0] == 69.0  # pure-match floor, u32 PML
    assert mod.HBM_PEAK_GBS == 8000.0
    # warm-up characters: enough for sigma^w >= runs
    assert mod.warmup_for(a, 253, 10**9) == 4 and mod.warmup_for(a, 4, 10**9) == 15


def test_only_the_cpu_baseline_touches_the_oracle():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    where = []
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            names = []
            if isinstance(node, ast.Import):
                names = [x.name for x in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            if any(n == "oracle" or n.startswith("oracle.") for n in names):
                where.append(fn.name)
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any("oracle" in ast.dump(n) for n in top)
    assert set(where) <= {"cpu_baseline", "run_c4_ms_doc", "run_real_bwt", "run_real_bwt_ms_doc", "run_long_reads_c5"}, where  # the legs' parity gates and the baseline
    assert "cpu_baseline" in where


def test_the_launcher_line_is_the_drivers(monkeypatch):
    mod, a = _bench(["--gpus", "2", "--steps", "3"])
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(mod.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    assert mod.respawn_under_torchrun(a) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
