"""The C ABI without Python: ``tests/c_abi/standalone.cpp`` includes ``include/pf_amd.h``, links ``libpfamd.so`` and the
HIP runtime, allocates device memory itself and runs SISR + Bootstrap and APF + LinearGaussianObservations on the
reference's AR(1) test model - the whole time loop behind one ``pf_filter_run`` call - against an exact Kalman filter
computed on the host (log-likelihood within 0.25, final mean within 0.01 at 65 536 particles) - and the theta-level entry
points (``pf_theta_ess / _resample / _fit``, ``pf_theta_step`` with its polled host slot) against the same arithmetic on the host."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_standalone_program_against_kalman(tmp_path):
    import __graft_entry__ as ge

    ge.build()
    exe = str(tmp_path / "standalone")
    lib_dir = os.path.join(ROOT, "pyfilter_amd")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{ROOT}/include",
           os.path.join(ROOT, "tests", "c_abi", "standalone.cpp"), "-o", exe, f"-L{lib_dir}", "-lpfamd", f"-L{rocm}/lib",
           "-lamdhip64", f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{rocm}/lib"]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    # (2 variants + the cluster route) x 2 columns x (ll, mean) + the line of the forced give-up
    assert "c-abi ok" in run.stdout and run.stdout.count("Kalman") == 13
    assert "cluster route (launch trace 10)" in run.stdout
    assert "cluster_patience = -1: status 1 -> re-issued with PF_ROUTE_PER_STEP" in run.stdout
    assert "theta level: ESS" in run.stdout
    assert "theta step: observation 3 polled from host memory" in run.stdout
