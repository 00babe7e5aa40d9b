"""bench.py's host-side pieces that need no GPU: how the rocprofv3 --pmc child runs' CSVs become `roofline.traffic`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def _write(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write('"Correlation_Id","Dispatch_Id","Agent_Id","Kernel_Name","Counter_Name","Counter_Value"\n')
        for i, (kernel, counter, value) in enumerate(rows):
            f.write('%d,%d,"Agent 4","%s","%s",%s\n' % (i, i, kernel, counter, value))


def test_counter_mean_takes_the_second_half_of_the_frame_loops_dispatches(tmp_path):
    k = "void spk2::k_frames<10, false, 8>(spk::FrameArgs, int, HIP_vector_type<double, 2u> const*, int, int)"
    rows = [(k, "FETCH_SIZE", 1000.0 + i) for i in range(10)]                    # warm-up half 1000..1004, counted half 1005..1009
    rows += [("void spk::k_synth_trinoise<12>(spk::SynthArgs)", "FETCH_SIZE", 9e9)]  # another kernel of the run: ignored
    rows += [(k, "WRITE_SIZE", 5.0)] * 10                                           # another counter: ignored
    _write(str(tmp_path / "host" / "123_counter_collection.csv"), rows)
    assert bench.pmc_counter_mean(str(tmp_path), "FETCH_SIZE") == 1007.0
    assert bench.pmc_counter_mean(str(tmp_path), "WRITE_SIZE") == 5.0


def test_counter_mean_refuses_a_run_that_is_too_short_or_empty(tmp_path):
    k = "void spk2::k_frames<10, false, 8>(spk::FrameArgs, int, HIP_vector_type<double, 2u> const*, int, int)"
    _write(str(tmp_path / "a" / "1_counter_collection.csv"), [(k, "FETCH_SIZE", 1.0)] * 7)
    assert bench.pmc_counter_mean(str(tmp_path), "FETCH_SIZE") is None
    assert bench.pmc_counter_mean(str(tmp_path / "nothing_here"), "FETCH_SIZE") is None
