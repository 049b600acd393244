"""CPU: every adapter template of include/vloam_hip/compat.hpp and every Create() factory of include/vloam_hip/factors.hpp instantiated with the
reference's own argument types — pcl::PointCloud<PointType>::Ptr, pcl::PointCloud<pcl::PointXYZ>, Eigen::Quaterniond / Vector3d, cv::Mat,
ceres::CostFunction*, tf2::Transform — as STAND-INS (tests/stubs/: PCL, Eigen, OpenCV, Ceres and tf2 do not exist in this image).  This is a
syntax / overload-resolution check of OUR headers with -DVLOAM_HIP_WITH_PCL -DVLOAM_HIP_WITH_OPENCV -DVLOAM_HIP_WITH_CERES, not an oracle and not a
build of the reference: the probe's façade (tests/cpp/ref_facade_probe.cpp) makes the calls of lidar_odometry_mapping.cpp:73-154 with their argument
lists, and tests/test_gpu_cpp_boundary.py::test_reference_typed_facade_on_the_gpu runs the same binary on the GPU against the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("compiler,std", [("g++", "c++14"), ("g++", "c++17"), ("/opt/rocm/lib/llvm/bin/clang++", "c++17")])
def test_adapters_compile_with_the_reference_types_and_selfcheck(tmp_path, vl, compiler, std):
    vl.lib()   # (the probe links against the built library)
    if not os.path.exists(compiler) and compiler.startswith("/"):
        pytest.skip("no clang++ at " + compiler)
    exe = tmp_path / "ref_probe"
    libdir = os.path.join(ROOT, "vloam-cmu-16833_amd")
    subprocess.check_call([compiler, "-std=" + std, "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "stubs"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "ref_facade_probe.cpp"), "-o", str(exe),
                           "-L", libdir, "-lvloam_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe), "--selfcheck"]).decode()
    assert "selfcheck OK" in out


def test_headers_stay_free_of_third_party_includes_by_default(tmp_path):
    """Without the VLOAM_HIP_WITH_* switches the headers must compile with nothing but the C++ standard library on the include path."""
    src = tmp_path / "plain.cpp"
    src.write_text('#include "vloam_hip/compat.hpp"\n#include "vloam_hip/factors.hpp"\nint main() { vloam::Transform t; return t.q[3] == 1.0 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
