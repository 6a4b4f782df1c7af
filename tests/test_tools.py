"""The measurement scripts under tools/ and the bench parse (they only run on a GPU box; a typo in one of them
would otherwise show up in the middle of a profiling session)."""
import ast
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tools_and_bench_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"),
                                                                     os.path.join(ROOT, "__graft_entry__.py")]
    assert len(files) > 10
    for f in files:
        with open(f) as fh:
            ast.parse(fh.read(), filename=f)


def test_profile_round_script_mentions_only_existing_tools():
    text = open(os.path.join(ROOT, "tools", "profile_round.sh")).read()
    for word in text.replace("/", " ").split():
        if word.endswith(".py") and not word.startswith("$"):
            assert any(os.path.exists(os.path.join(ROOT, d, word)) for d in ("tools", "", "tests")), word
