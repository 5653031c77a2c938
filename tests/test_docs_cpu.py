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


# ---------------------------------------------------------------------------------------------------------------------
# Numbers in prose come from the records.
#
# README.md and DESIGN.md carry their figures in table rows that NAME the record and the key they quote:
#     | config 2, frames/s | 2 260 | `profiles/r06_bench_cfg2.json` : `value` |
# Every such row is held to its record (1 % / 0.005 for fractions), the record has to be the NEWEST round's file of that
# name, and every "<number> frames/s" anywhere else in the two files has to be a figure some newest bench record holds
# (3 %: box-to-box spread) - a sentence cannot keep an old number alive.
import json

ROW = re.compile(r"^\|(?P<what>[^|]*)\|(?P<val>[^|]*)\|\s*`(?P<rec>profiles/[A-Za-z0-9_./-]+\.json)`\s*:\s*`(?P<key>[A-Za-z0-9_.\[\]]+)`\s*\|\s*$")
NUM = r"\d{1,3}(?:[   ,]\d{3})+(?:\.\d+)?|\d+(?:\.\d+)?"
FPS = re.compile(rf"(?P<a>{NUM})(?:\s*[–-]\s*(?P<b>{NUM}))?\s*frames/s")


def _num(txt):
    return float(re.sub(r"[   ,]", "", txt))


def _record(path):
    """a bench line (one JSON object on the last non-empty line) or a plain JSON file"""
    text = open(os.path.join(ROOT, path)).read().strip()
    try:
        return json.loads(text)
    except ValueError:
        return json.loads(text.splitlines()[-1])


def _get(rec, key):
    cur = rec
    for part in key.split("."):
        m = re.fullmatch(r"([A-Za-z0-9_]+)\[(\d+)\]", part)
        cur = cur[m.group(1)][int(m.group(2))] if m else cur[part]
    return cur


def _newest(path):
    """profiles/r05_bench_cfg2.json -> the same name of the newest round that has it"""
    d, base = os.path.split(path)
    m = re.fullmatch(r"r(\d+)_(.*)", base)
    assert m, path
    rounds = sorted(glob.glob(os.path.join(ROOT, d, f"r[0-9][0-9]_{m.group(2)}")))
    return os.path.join(d, os.path.basename(rounds[-1]))


def _rows(doc):
    for line in open(os.path.join(ROOT, doc)):
        m = ROW.match(line.rstrip("\n"))
        if m:
            yield m


def test_number_rows_agree_with_their_records():
    checked, wrong = 0, []
    for doc in ("README.md", "DESIGN.md"):
        for m in _rows(doc):
            rec_path, key = m.group("rec"), m.group("key")
            if _newest(rec_path) != rec_path:
                wrong.append((doc, m.group("what").strip(), f"{rec_path} is not the newest record ({_newest(rec_path)})"))
                continue
            want = float(_get(_record(rec_path), key))
            got = _num(re.search(NUM, m.group("val")).group(0))
            tol = 0.005 if abs(want) < 1.5 else 0.01 * abs(want)
            checked += 1
            if abs(got - want) > tol:
                wrong.append((doc, m.group("what").strip(), got, want))
    assert not wrong, wrong
    assert checked >= 20, checked                        # the tables exist


def test_frames_per_second_in_prose_is_a_recorded_figure():
    allowed = []
    for path in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_cfg*.json")):
        rel = os.path.relpath(path, ROOT)
        if _newest(rel) != rel:
            continue
        rec = _record(rel)
        allowed.append(rec["value"])
        for sub in ("split_bf16x3", "cpu_baseline"):
            if sub in rec:
                allowed.append(rec[sub]["value"])
                for leaf in ("single_thread", "all_threads"):
                    if leaf in rec[sub]:
                        allowed.append(rec[sub][leaf]["value"])
    assert allowed
    stray = []
    for doc in ("README.md", "DESIGN.md"):
        for n, line in enumerate(open(os.path.join(ROOT, doc)), 1):
            if ROW.match(line.rstrip("\n")):
                continue
            for m in FPS.finditer(line):
                for txt in (m.group("a"), m.group("b")):
                    if txt is None:
                        continue
                    v = _num(txt)
                    if not any(abs(v - a) <= 0.03 * max(abs(a), 1e-9) + 0.0006 for a in allowed):
                        stray.append((doc, n, txt))
    assert not stray, stray


def test_design_config_table_is_the_generated_one():
    """DESIGN.md section 6's per-configuration table is tools/doc_tables.py's output for the newest records, verbatim."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("doc_tables", os.path.join(ROOT, "tools", "doc_tables.py"))
    dt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dt)
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    a, b = text.index(dt.BEGIN) + len(dt.BEGIN), text.index(dt.END)
    have = [ln for ln in text[a:b].strip("\n").split("\n")]
    assert have == dt.table(), "run: python tools/doc_tables.py --write"
