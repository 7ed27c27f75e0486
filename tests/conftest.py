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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
