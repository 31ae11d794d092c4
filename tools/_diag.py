"""Diagnostics build of libkmgpu.so (-DKM_DIAGNOSTICS [+ extra defines]): the timing ablations
(KM_DEBUG_FLAGS, KM_DEBUG_DELIVER, KM_DELIVER_ZEROCOPY, KM_DFS_REPLAY) exist only there — their results
are invalid, and the product library ignores them.  build() compiles it into the temp directory and
points KM_LIBRARY at it; call it before importing km_amd.lib."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "km_amd", "csrc")


def build(extra=(), tag="diag"):
    so = os.path.join(tempfile.gettempdir(), "libkmgpu_%s.so" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-ffp-contract=off", "-pthread", "-DKM_DIAGNOSTICS"] + list(extra) +
                          ["-o", so, os.path.join(CSRC, "kmgpu.hip"), os.path.join(CSRC, "jf_reader.cpp"),
                           os.path.join(CSRC, "report.cpp")], cwd=ROOT)
    os.environ["KM_LIBRARY"] = so
    return so


if __name__ == "__main__":
    print(build())
