"""bench.py's own launcher and its line guard (no GPU needed): `python bench.py --gpus N` starts N ranks itself, a mismatch
between --gpus and the launcher's WORLD_SIZE fails loudly, and a leg that hangs costs that leg -- never the line -- while the
process still ends with a failure code.  (The N > 1 line itself -- cpu_baseline, n1_reference, over_n1 -- is checked on the GPU box by
tests/test_gpu_scale.py::test_bench_two_rank_rehearsal_line_is_complete.)"""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_launch_starts_one_child_per_rank_and_relays_the_exit_code(tmp_path):
    stub = tmp_path / "stub.py"
    stub.write_text(textwrap.dedent("""
        import os, sys
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        if r == 0:
            print('{"n_gpus": %d, "argv": "%s"}' % (w, " ".join(sys.argv[1:])), flush=True)
        sys.exit(int(os.environ.get("STUB_FAIL_RANK", "-1")) == r and 5 or 0)
    """))
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; sys.exit(bench.launch(3, ['--gpus', '3', '--steps', '2'], script={str(stub)!r}))"
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"n_gpus": 3, "argv": "--gpus 3 --steps 2"}
    r = subprocess.run([sys.executable, "-c", code], env=_env(STUB_FAIL_RANK="2"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 5


def test_importing_bench_does_not_load_torch_or_the_library():
    """the launcher parent must not touch the GPU: importing bench pulls in neither torch nor liblinne_amd"""
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; assert 'torch' not in sys.modules and 'linne_amd' not in sys.modules"
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr


def test_gpus_must_match_the_launchers_world_size():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 8" in r.stderr


def test_a_hung_leg_costs_the_leg_not_the_line_and_fails_the_run():
    code = textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        g = bench.Guard(0)
        g.line = {{"value": 1.0, "transports": {{}}}}
        g.line["a"] = g.run("a", 5, lambda: {{"fine": True}})
        g.line["b"] = g.run("b", 5, lambda: 1 / 0)
        g.run("transports.rccl", 0.5, lambda: time.sleep(60), on_timeout=lambda rec: g.line["transports"].__setitem__("rccl", rec))
        print("not reached")
    """)
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=60)
    assert r.returncode == 3, r.stderr         # the line is printed, and a hang is still a failure of the run
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] == 1.0 and line["a"] == {"fine": True} and "ZeroDivisionError" in line["b"]["error"]
    assert "timed out" in line["transports"]["rccl"]["error"] and line["legs_timed_out"] == ["transports.rccl"]


def test_a_parity_failure_before_a_hang_keeps_its_own_exit_code():
    code = textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        g = bench.Guard(0)
        g.line = {{"value": 1.0}}
        g.exit_code = 2
        g.run("x", 0.3, lambda: time.sleep(60))
    """)
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=60)
    assert r.returncode == 2, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["legs_timed_out"] == ["x"]
