"""The PlusCal front-end must REFUSE what it cannot read, never crash: the algorithms of specs/pluscal/ with random token-level damage
(a token deleted, duplicated, swapped, replaced by punctuation / a keyword) go through parser, procedure expansion, record flattening,
translator and compiler in a child process — every outcome but a normal exit (a message, or a program) fails the test.  (Round 5 found a
segmentation fault of the translator on a well-formed input this way of looking would not have produced — a uniprocess algorithm whose
procedure has a parameter, tests/test_pcal.py — and then ran 800 mutations under ASan / UBSan: none crashed; this keeps it so.)"""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CHILD = r'''
import glob, random, re, sys
sys.path.insert(0, sys.argv[1] + "/tests")
import helpers
from test_pcal import strip_translation
CONST = {"N": 2, "K": 3, "RM": 2, "M": 6, "P": 2, "Rounds": 1, "Items": 3, "MaxQ": 2, "Consumers": 1, "Workers": 2, "Torn": 0, "Racy": 0, "Counted": 1,
         "Eager": 0, "Hasty": 0, "Bound": 3}
specs = sorted(glob.glob(sys.argv[1] + "/specs/pluscal/*.tla"))
r = random.Random(int(sys.argv[2]))
ok = refused = 0
for it in range(int(sys.argv[3])):
    text = strip_translation(open(r.choice(specs)).read())
    a = text.find("--algorithm")
    e = text.find("end algorithm") if "end algorithm" in text else text.find("*)", a)
    toks = re.findall(r"\s+|\w+|\\[a-zA-Z]+|[^\w\s]", text[a:e])
    kind = r.randrange(4)
    for _ in range(r.randrange(1, 3)):
        i = r.randrange(len(toks))
        if kind == 0:
            del toks[i]
        elif kind == 1:
            toks.insert(i, r.choice(toks))
        elif kind == 2:
            j = r.randrange(len(toks))
            toks[i], toks[j] = toks[j], toks[i]
        else:
            toks[i] = r.choice([";", ":=", "(", ")", "[", "]", "<<", ">>", "{", "}", "end", "if", "with", "||", ".", ",", "\\in", "self", "0", "|->"])
    mut = text[:a] + "".join(toks) + text[e:]
    print(it, flush=True)
    try:
        helpers.pcal_translate(mut)
        helpers.ShimProgram(mut, [], CONST).close()
        ok += 1
    except RuntimeError as ex:
        assert str(ex).strip()
        refused += 1
print("done", ok, refused)
'''


@pytest.mark.parametrize("seed", range(4))
def test_damaged_algorithms_are_refused_not_crashed_on(seed):
    p = subprocess.run([sys.executable, "-c", CHILD, str(ROOT), str(seed), "120"], capture_output=True, text=True, timeout=600)
    last = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
    assert p.returncode == 0 and last.startswith("done"), f"seed {seed}: the front-end died on mutation {last} (rc {p.returncode}): {p.stderr[-400:]}"
    ok, refused = map(int, last.split()[1:])
    assert refused > 40 and ok + refused == 120


CFG_CHILD = r'''
import glob, random, re, sys
sys.path.insert(0, sys.argv[1])
import tla_rust_amd as amd
cfgs = sorted(glob.glob(sys.argv[1] + "/specs/**/*.cfg", recursive=True)) + sorted(glob.glob("/root/reference/examples/**/*.cfg", recursive=True))
r = random.Random(int(sys.argv[2]))
ok = refused = 0
for it in range(int(sys.argv[3])):
    path = r.choice(cfgs)
    toks = re.findall(r"\s+|\w+|[^\w\s]", open(path).read())
    for _ in range(r.randrange(1, 4)):
        i = r.randrange(len(toks))
        k = r.randrange(4)
        if k == 0:
            del toks[i]
        elif k == 1:
            toks.insert(i, r.choice(toks))
        elif k == 2:
            toks[i] = r.choice(["=", "{", "}", "<-", "[", "]", "CONSTANT", "INVARIANT", "SPECIFICATION", "CHECK_DEADLOCK", "x", "3", ",", '"', "(*", "\\*", "-"])
        else:
            j = r.randrange(len(toks))
            toks[i], toks[j] = toks[j], toks[i]
    text = "".join(toks)
    print(it, flush=True)
    try:
        amd.cfg_parse(text)
        ok += 1
        for module in ("raft", "MCraft", "pcal_intro", "atomic_add", "MCssi", "MCPaxos"):
            try:
                amd.spec_resolve(module, text)
            except amd.McError as ex:
                assert str(ex).strip()
    except amd.McError as ex:
        assert str(ex).strip()
        refused += 1
print("done", ok, refused)
'''


@pytest.mark.parametrize("seed", range(2))
def test_damaged_configuration_files_are_refused_not_crashed_on(seed):
    """the cfg parser and the spec registry (frontend.cpp: mc_cfg_parse, mc_spec_resolve) on token-level mutations of every cfg of specs/ and
    of the reference tree: a message or a result, never a crash"""
    p = subprocess.run([sys.executable, "-c", CFG_CHILD, str(ROOT), str(seed), "300"], capture_output=True, text=True, timeout=600)
    last = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
    assert p.returncode == 0 and last.startswith("done"), f"seed {seed}: the cfg front-end died on mutation {last} (rc {p.returncode}): {p.stderr[-400:]}"
    ok, refused = map(int, last.split()[1:])
    assert refused > 30 and ok > 30
