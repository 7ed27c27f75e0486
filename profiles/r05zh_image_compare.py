"""After the last whole-suite GPU run (r05zb, commit f079a81) the PlusCal front-end kept changing on the HOST side (LET, CASE, DOMAIN, set filters,
CHOOSE over sets, operators over whole variables ...).  This compares the compiled program IMAGES (what the GPU interprets) of every program
tests/test_gpu_pcal.py and tests/test_gpu_zz_ms_queue.py run — the spec cases, 60 fuzz and 40 random algorithms — between a build of that
commit (git worktree + g++ of tests/_shim/shim.cpp into /tmp/w/libshim_old.so) and the current tree: 141 of 141 bit-identical at the end of
round 5.  (The channel / soup / roadmap cases of tests/test_gpu_zz_channels.py were re-run on the GPU instead: r05zg.)"""
import sys, ctypes as C, os
sys.path.insert(0,'/root/repo/tests')
import helpers
from test_pcal import CASES, CHANNEL_STEMS
from test_pcal_fuzz import Gen
from test_pcal_random import Gen as RGen
new = C.CDLL(str(helpers.build_shim())); old = C.CDLL('/tmp/w/libshim_old.so')
def image(lib, text, invs, consts):
    lib.shim_program_compile2.restype = C.c_void_p
    lib.shim_program_compile2.argtypes = [C.c_char_p]*4
    lib.shim_program_free.argtypes = [C.c_void_p]
    cs = ",".join(f"{k}={int(v)}" for k, v in (consts or {}).items())
    h = lib.shim_program_compile2(text.encode(), ",".join(invs).encode(), cs.encode(), b"")
    if not h: return None
    b = C.cast(h + 8, C.POINTER(C.c_void_p))[0]; e = C.cast(h + 16, C.POINTER(C.c_void_p))[0]
    n = (e - b) // 4
    img = list(C.cast(b, C.POINTER(C.c_int32))[0:n])
    lib.shim_program_free(h)
    return img
same = diff = 0
progs = [(p.read_text(), invs, consts, p.stem + str(consts)) for p, invs, consts in CASES if p.stem not in CHANNEL_STEMS]
progs += [(Gen(s).program()[0], Gen(s).program()[1], {}, f"fuzz{s}") for s in range(60)]
progs += [(RGen(s).module(f"rnd{s}"), ["Small"], {}, f"rnd{s}") for s in range(1000, 1040)]
for text, invs, consts, tag in progs:
    a = image(old, text, invs, consts); b = image(new, text, invs, consts)
    if a == b: same += 1
    else:
        diff += 1; print("DIFF", tag, None if a is None else len(a), None if b is None else len(b))
print("same", same, "diff", diff)
