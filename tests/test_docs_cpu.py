"""Doc drift guards: DESIGN.md stays readable in one sitting, and every repository path that
README / DESIGN / INTEGRATION cite exists."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CITED = re.compile(r"`((?:profiles|tests|tools|diffroll_amd|include|oracle)/[A-Za-z0-9_./*{}\[\],<>-]+)`")


def _paths(doc):
    text = open(os.path.join(ROOT, doc)).read()
    for m in CITED.finditer(text):
        p = m.group(1).rstrip(".,")
        p = p.split("::")[0]
        if "<" in p or ">" in p:                         # placeholders: profiles/r<NN>_...
            continue
        if p.startswith("oracle/_ref"):                  # named to say that it does NOT exist (a pure-Python reference)
            continue
        yield p


def _exists(p):
    if "{" in p:                                         # r05_bench_cfg{1..7}.json, r05_kernel_stats_cfg{3,4,5}.txt
        head, rest = p.split("{", 1)
        inner, tail = rest.split("}", 1)
        if ".." in inner:
            a, b = inner.split("..")
            items = [str(i) for i in range(int(a), int(b) + 1)]
        else:
            items = inner.split(",")
        return all(_exists(head + it + tail) for it in items)
    full = os.path.join(ROOT, p)
    if "*" in p:
        return bool(glob.glob(full))
    return os.path.exists(full)


def test_design_md_is_short_and_history_exists():
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) <= 35 * 1024
    assert os.path.exists(os.path.join(ROOT, "DESIGN_HISTORY.md"))


def test_cited_paths_exist():
    missing = []
    for doc in ("README.md", "DESIGN.md", "INTEGRATION.md"):
        for p in _paths(doc):
            if not _exists(p):
                missing.append((doc, p))
    assert not missing, missing
