import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    if os.environ.get("MPHIP_POISON_EMPTY") == "1":
        # hunt for reads of uninitialised device memory: every torch.empty() comes back filled with NaN (floats) / max int,
        # so a kernel that reads a workspace word, range-descriptor slot or output element nobody wrote turns a test red
        import torch

        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_c():
    """The plain-C oracle (oracle/hotpath_c.c), built on demand with gcc."""
    import subprocess

    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    from oracle import hotpath_c

    return hotpath_c


@pytest.fixture
def c_abi_exe(tmp_path):
    """tests/c_abi/c_abi_smoke.c built with gcc as C99 against include/mphip.h + libmphip.so (+ the HIP runtime for
    device memory).  Compiling/linking needs no GPU; running it does."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "megaportrait-hack_amd")
    exe = os.path.join(str(tmp_path), "c_abi_smoke")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(root, "include"),
           "-I", "/opt/rocm/include", os.path.join(root, "tests", "c_abi", "c_abi_smoke.c"), "-o", exe, "-L", pkg, "-lmphip",
           "-L", "/opt/rocm/lib", "-lamdhip64", "-lm", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.fixture
def plan_c_exe(tmp_path, oracle_c):
    """tests/c_abi/plan_smoke.c (C99, gcc) against include/mphip.h + libmphip.so, checked against the plain-C oracle
    (oracle/libmphip_oracle.so: test infrastructure) — compiling/linking needs no GPU; running it does."""
    import subprocess

    pkg = os.path.join(ROOT, "megaportrait-hack_amd")
    orc = os.path.join(ROOT, "oracle")
    exe = os.path.join(str(tmp_path), "plan_smoke")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "plan_smoke.c"), "-o", exe, "-L", pkg, "-lmphip",
           "-L", orc, "-lmphip_oracle", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm", "-fopenmp", f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{orc}",
           "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe
