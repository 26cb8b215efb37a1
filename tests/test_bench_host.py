"""-m "not gpu": host logic of bench.py that does not need a GPU — the parsing of the rocprofv3 --pmc counter
files behind roofline.traffic (with a stand-in for rocprofv3 that writes the CSV a real run produces), and the
kernel-name shortening."""
import os
import stat
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_FAKE = textwrap.dedent('''\
    #!/usr/bin/env python3
    # stand-in for rocprofv3: rocprofv3 --pmc COUNTER --output-format csv -d DIR -- python bench.py ...
    import os, sys
    a = sys.argv[1:]
    counter, d = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
    os.makedirs(os.path.join(d, "host"), exist_ok=True)
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"]
    conv = '"void conv_igemm_kernel<128, 128, 2, 2, false, 2>(ConvArgs, int)"'
    vals = {"FETCH_SIZE": (1000.0, 3000.0), "WRITE_SIZE": (500.0, 700.0)}[counter]   # KiB per dispatch
    for i, v in enumerate(vals):
        rows.append("1,%d,0,1,1,1,256,7,%s,256,0,0,64,0,32,%s,%f,0,1" % (i + 1, conv, counter, v))
    rows.append('1,9,0,1,1,1,256,8,"__amd_rocclr_fillBufferAligned",256,0,0,64,0,32,%s,999999,0,1' % counter)
    rows.append('1,10,0,1,1,1,256,9,"bn_fold_kernel(float*, FoldDesc const*, float)",256,0,0,64,0,32,%s,%f,0,1' % (counter, 10.0))
    open(os.path.join(d, "host", "1_counter_collection.csv"), "w").write("\\n".join(rows) + "\\n")
    ''')


def test_short_kernel_name():
    import bench
    assert bench._short_kernel_name("void conv_igemm_kernel<128, 128, 2, 2, false, 2>(ConvArgs, int)") == \
        "void conv_igemm_kernel<128, 128, 2, 2, false, 2>"
    assert bench._short_kernel_name("(anonymous namespace)::hist_kernel(DetArgs, void*, int, int)") == \
        "(anonymous namespace)::hist_kernel"
    assert bench._short_kernel_name("__amd_rocclr_copyBuffer") == "__amd_rocclr_copyBuffer"


def test_traffic_from_counter_files(tmp_path, monkeypatch):
    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text(_FAKE)
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setattr(bench, "ROCPROF", str(fake))
    out, note = bench.measure_hbm_traffic(["--steps", "1"], steps_run=2)
    assert note is None
    k = out["void conv_igemm_kernel<128, 128, 2, 2, false, 2>"]
    assert k["launches"] == 2
    assert k["fetch_bytes"] == 2 * (1000.0 + 3000.0) * 1024 / 2          # KiB -> bytes, x2 (gfx950), mean per launch
    assert k["write_bytes"] == (500.0 + 700.0) * 1024 / 2
    assert k["hbm_bytes"] == k["fetch_bytes"] + k["write_bytes"]
    # per step: library kernels only (no runtime fill / copy kernels), dispatches / steps_run
    want = (k["hbm_bytes"] * 2 + (2 * 10.0 + 10.0) * 1024) / 2
    assert abs(out["_per_step"] - want) < 1e-6


def test_traffic_reports_a_missing_profiler(monkeypatch):
    import bench
    monkeypatch.setattr(bench, "ROCPROF", "/nonexistent/rocprofv3")
    out, note = bench.measure_hbm_traffic([], 1)
    assert out is None and "not found" in note


def test_forward_tile_key_matches_both_template_signatures():
    import bench
    names = ["void conv_igemm_kernel<128, 128, 2, 2, true, 2>", "void conv_igemm_kernel<64, 64, 2, 2, false, 4>",
             "void conv_igemm_kernel<64, 64, 2, 2, false, 2>", "void conv_igemm_kernel<128, 128, 2, 2, false, 2>",
             "void conv_igemm_kernel<128, 64, 2, 2, false>", "sgd_kernel"]
    assert bench._forward_tile_key(names, "128", "128") == "void conv_igemm_kernel<128, 128, 2, 2, false, 2>"
    assert bench._forward_tile_key(names, "64", "64") == "void conv_igemm_kernel<64, 64, 2, 2, false, 2>"
    assert bench._forward_tile_key(names, "128", "64") == "void conv_igemm_kernel<128, 64, 2, 2, false>"
    assert bench._forward_tile_key(names, "128", "32") is None
    # seven-argument names: the stream-K instance is a kernel of its own
    names = ["void conv_igemm_kernel<128, 128, 2, 2, false, 2, true>", "void conv_igemm_kernel<128, 128, 2, 2, false, 2, false>"]
    assert bench._forward_tile_key(names, "128", "128") == names[1]
    assert bench._forward_tile_key(names, "128", "128", streamk=True) == names[0]


def test_no_nested_profiler():
    """bench.py inside rocprofv3 must not start its own rocprofv3 children, and the children it does start must not
    inherit profiler settings."""
    import bench
    assert not bench.under_profiler({"PATH": "/usr/bin", "LD_PRELOAD": "/lib/libfoo.so"})
    assert bench.under_profiler({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})
    assert bench.under_profiler({"ROCP_TOOL_LIBRARIES": "x"}) and bench.under_profiler({"ROCPROF_OUTPUT_PATH": "/tmp"})
    env = bench.clean_child_env({"PATH": "p", "ROCP_TOOL_LIBRARIES": "x", "ROCPROFILER_REGISTER_FORCE_LOAD": "1",
                                 "HSA_TOOLS_LIB": "libx.so", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                                 "LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so:/lib/libfoo.so"})
    assert env == {"PATH": "p", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "LD_PRELOAD": "/lib/libfoo.so"}
    assert "LD_PRELOAD" not in bench.clean_child_env({"LD_PRELOAD": "/x/librocprofiler-sdk-tool.so"})


def test_traffic_is_skipped_inside_a_profiler(monkeypatch, tmp_path):
    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text(_FAKE)
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setattr(bench, "ROCPROF", str(fake))
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    out, note = bench.measure_hbm_traffic(["--steps", "1"], steps_run=2)
    assert out is None and "profiled" in note


def test_ipc_mode_is_set_before_torch_whatever_the_launcher():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC; RCCL across processes needs it on this host driver)
    used to be set only by the repo's own launcher — a rank started by `python -m torch.distributed.run bench.py` never
    had it.  bench.py and the package now set it themselves, before torch is imported (the HSA runtime reads it when the
    first GPU call initialises it)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    code = ("import sys, os; sys.path.insert(0, %r)\n"
            "import %s\n"
            "assert os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0', 'not set'\n"
            "print('torch' in sys.modules)")
    for mod in ("bench", "videoyolo_amd"):
        p = subprocess.run([sys.executable, "-c", code % (ROOT, mod)], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
    # bench.py: set at module level, before anything could have imported torch
    p = subprocess.run([sys.executable, "-c", code % (ROOT, "bench")], env=env, capture_output=True, text=True, timeout=300)
    assert p.stdout.strip() == "False", "importing bench.py pulled torch in before main(): the setdefault must stay ahead of it"
    # an explicit setting of the caller wins
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "1"
    p = subprocess.run([sys.executable, "-c", "import sys, os; sys.path.insert(0, %r); import bench; print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])" % ROOT],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.stdout.strip() == "1"


_DOG = textwrap.dedent('''\
    import sys, time
    sys.path.insert(0, %r)
    import bench
    result = {"value": 123.0, "also_416": {"frames_per_s": 1.0}} if sys.argv[1] == "with_headline" else {}
    dog = bench.LegWatchdog(0, 1.0, result)
    dog.arm("also_train416")
    dog.disarm()
    time.sleep(1.6)               # a disarmed budget never fires
    dog.arm("also_syncbn608")
    time.sleep(30)                # "a collective whose peer never arrives"
    print("NOT REACHED")
    ''')


def test_leg_watchdog_prints_the_line_and_ends_the_run(tmp_path):
    """A leg that exceeds its wall-clock budget: rank 0 prints the contract line with everything measured before it and the
    leg as {"timeout": true}; exit code 0 when the headline value is in the line, 4 when nothing was measured."""
    import json
    import subprocess
    import time
    script = tmp_path / "dog.py"
    script.write_text(_DOG % ROOT)
    for mode, rc in (("with_headline", 0), ("nothing", 4)):
        t0 = time.time()
        p = subprocess.run([sys.executable, str(script), mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        assert p.returncode == rc and time.time() - t0 < 25, (p.returncode, p.stderr.decode()[-500:])
        out = p.stdout.decode()
        assert "NOT REACHED" not in out
        line = json.loads(out.strip().splitlines()[-1])
        assert line["also_syncbn608"] == {"timeout": True, "budget_s": 1.0} and line["aborted_after_timeout_of"] == "also_syncbn608"
        assert "also_train416" not in line            # it was disarmed in time
        if mode == "with_headline":
            assert line["value"] == 123.0 and line["also_416"] == {"frames_per_s": 1.0}
        assert "exceeded its wall-clock budget" in p.stderr.decode()


def test_leg_watchdog_accounts_wall_clock_per_leg():
    """`wall_s` of the line: seconds per leg between arm / arm / disarm (a leg armed twice accumulates), and since the start."""
    import time
    import bench
    dog = bench.LegWatchdog(0, 0, {})       # budget 0: no thread, accounting only
    dog.arm("headline")
    time.sleep(0.05)
    dog.arm("also_416")
    time.sleep(0.02)
    dog.disarm()
    time.sleep(0.02)                        # between legs: counted in since_start only
    dog.arm("headline")
    time.sleep(0.03)
    dog.disarm()
    w = dog.wall_report()
    assert set(w["legs"]) == {"headline", "also_416"}
    assert 0.07 <= w["legs"]["headline"] <= 0.5 and 0.01 <= w["legs"]["also_416"] <= 0.3
    assert w["since_start"] >= w["legs"]["headline"] + w["legs"]["also_416"]


def test_device_under_load_samples_while_the_step_runs():
    """`device_under_load` of the line: medians of the clock / power samples taken while the extra steps run; an error entry,
    not an exception, on a runtime without the queries."""
    import time
    import types
    import bench

    class Cuda(object):
        def __init__(self, ok=True):
            self.ok, self.n = ok, 0

        def clock_rate(self, i):
            if not self.ok:
                raise RuntimeError("amdsmi is not available")
            self.n += 1
            return 2300 + self.n % 3

        def power_draw(self, i):
            return 900

        def temperature(self, i):
            return 60

        def synchronize(self, dev):
            pass

    dev = types.SimpleNamespace(index=0)
    r = bench.device_under_load(types.SimpleNamespace(cuda=Cuda()), dev, lambda: time.sleep(0.03), steps=3, period_s=0.005)
    assert r["samples"] >= 5 and 2300 <= r["sclk_mhz_median"] <= 2302 and r["power_median"] == 900 and r["temperature_c_max"] == 60
    r = bench.device_under_load(types.SimpleNamespace(cuda=Cuda(ok=False)), dev, lambda: None)
    assert "amdsmi is not available" in r["error"]
