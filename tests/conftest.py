import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


# The MC wrappers of specs/ EXTEND modules of the reference (raft.tla, the snapshot-isolation specs), which the front-end
# verifies against the text its lowerings were written for.  In the build container they are read where they lie; the GPU
# box has no /root/reference, so there the built-in lowerings are accepted unverified (the report says so).
import os  # noqa: E402
_REF = Path("/root/reference/examples")
if _REF.exists():
    os.environ.setdefault("TLA_PATH", str(_REF))
else:
    os.environ.setdefault("TLAMC_UNVERIFIED", "1")


def _cpu_workers(config):
    """The CPU suite (`-m "not gpu"`, ~20 minutes of single-core work: oracle runs, evaluator fuzzing, multi-process gloo runs) is
    spread over the host's cores with pytest-xdist when nobody asked for something else: plain `python -m pytest tests/ -x -q -m
    "not gpu"` then takes a few minutes.  Never on a GPU box (one process owns the device; the `-m gpu` tests run one after the
    other), never when -n / --dist were given, and TLAMC_TEST_WORKERS=0 (or 1) turns it off."""
    # a worker is itself a pytest process that runs this hook (xdist/remote.py calls pytest_cmdline_main in it): it must never
    # become a controller of its own workers
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get("TLAMC_TEST_NO_SPREAD"):
        return 0
    if Path("/dev/kfd").exists() or not config.pluginmanager.hasplugin("xdist"):
        return 0
    if getattr(config.option, "numprocesses", None) is not None or getattr(config.option, "collectonly", False) or config.getoption("usepdb", False):
        return 0
    want = os.environ.get("TLAMC_TEST_WORKERS")
    n = int(want) if want is not None else min(8, os.cpu_count() or 1)
    return n if n > 1 else 0


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    n = _cpu_workers(config)
    if n:
        os.environ["TLAMC_TEST_NO_SPREAD"] = "1"  # inherited by every process this run starts (workers, torchrun children, nested pytest)
        config.option.numprocesses = n
        config.option.dist = "worksteal"  # a handful of tests take 30-80 s: idle workers take over what a busy one still has queued


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if getattr(config.option, "numprocesses", None) and not hasattr(config, "workerinput"):
        # the controller builds what the workers will load (each builder is a no-op when its library is fresh), so that
        # eight workers do not start eight compilers on the same sources
        import helpers
        helpers.build_oracle()
        helpers.build_shim()
        helpers.build_tlaeval_door()
        import tla_rust_amd.build as b
        b.build()


@pytest.fixture(scope="session")
def oracle():
    import helpers
    helpers.build_oracle()
    return helpers


@pytest.fixture(scope="session")
def shim():
    import helpers
    helpers.build_shim()
    return helpers
