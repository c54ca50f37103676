"""tools/pmc_traffic.py: which dispatches of a counter pass belong to the timed step, and the per-kernel read factors."""
import csv
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tool():
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(ROOT, "tools", "pmc_traffic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def write_pass(path, counter, kernels):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for i, (name, value) in enumerate(kernels, 1):
            w.writerow({"Dispatch_Id": i, "Kernel_Name": name, "Counter_Name": counter, "Counter_Value": value})


def test_timed_step_leaves_out_the_counted_steps_trailing_sum(tmp_path):
    tool = load_tool()
    frame = lambda counted: [  # noqa: E731
        ("__amd_rocclr_fillBufferAligned", 1.0),
        (f"void rt_render_stored_kernel<1, {counted}, false, false>(render_args)", 10.0),
        ("rt_probe_sum_kernel(unsigned int)", 2.0),
        (f"void rt_render_stored_kernel<1, {counted}, false, false>(render_args)", 100.0),
        ("rt_sum_split_kernel(unsigned int const*)", 4000.0),
    ]
    path = tmp_path / "fetch_counter_collection.csv"
    write_pass(path, "FETCH_SIZE", frame("true") + frame("false"))
    rows = tool.timed_step_rows(str(path), "FETCH_SIZE")
    assert [r["Kernel_Name"].split("(")[0] for r in rows] == [k.split("(")[0] for k, _ in frame("false")]
    assert sum(float(r["Counter_Value"]) for r in rows) == 4113.0
    # a closest-hit batch: the counted trace is followed by nothing of its own
    batch = [("void rt_trace_kernel<4, true>(trace_args)", 5.0), ("__amd_rocclr_fillBufferAligned", 1.0),
             ("void rt_trace_kernel<4, false>(trace_args)", 7.0)]
    write_pass(path, "FETCH_SIZE", batch)
    assert [float(r["Counter_Value"]) for r in tool.timed_step_rows(str(path), "FETCH_SIZE")] == [1.0, 7.0]


def test_read_factors():
    tool = load_tool()
    assert tool.read_factor("rt_trace_kernel<4, false>") == 1.0          # 64-byte node gathers: counted exactly
    assert tool.read_factor("rt_sum_split_kernel") == 2.0                # streamed reads: the counter shows half
    assert tool.read_factor("rt_render_stored_kernel<1, false, false, false>") == 2.0


def test_c5_shares_are_strips_unless_asked_otherwise():
    """bench.py --config c5 on N ranks: interleaved strip sets of all samples each (rows gathered, nothing summed);
    --sample-groups G pairs ranks up on larger strip sets with a G-th of the samples each."""
    import types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pixels, lanes = 3840 * 2160, 256 * 1024
    assert [bench.c5_sample_groups(types.SimpleNamespace(sample_groups=0), n, pixels, lanes) for n in (1, 2, 4, 8)] == [1, 1, 1, 1]
    assert bench.c5_sample_groups(types.SimpleNamespace(sample_groups=4), 8, pixels, lanes) == 4
