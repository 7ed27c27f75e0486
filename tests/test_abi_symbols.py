"""libtlamc.so loads and exports every symbol include/tlamc.h declares (no compute calls: CPU only)."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_every_declared_symbol_is_exported():
    import tla_rust_amd.build as b
    so = b.build()
    lib = ctypes.CDLL(str(so))
    header = (ROOT / "include" / "tlamc.h").read_text()
    names = set(re.findall(r"\b(mc_[a-z_]+)\s*\(", header))
    assert len(names) >= 25
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_no_cpu_fallback_without_a_device():
    """The product path must fail loudly without a GPU (and never route through the oracle)."""
    import tla_rust_amd as amd
    if amd.device_count() > 0:
        return
    try:
        amd.Engine("atomic_add", [3])
    except amd.McError as e:
        assert e.code == -2 and "no CPU fallback" in str(e)
    else:
        raise AssertionError("engine creation succeeded without a HIP device")


def test_product_does_not_reference_the_oracle():
    for p in list((ROOT / "tla_rust_amd").rglob("*.py")) + list((ROOT / "tla_rust_amd" / "csrc").glob("*")):
        if p.is_file() and p.suffix in (".py", ".h", ".hip", ".cpp"):
            assert "oracle/" not in p.read_text().replace("oracle/spec_raft.c:raft_print", ""), p
