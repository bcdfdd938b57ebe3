"""CPU: the figures DESIGN.md (sections 4.1 and 6) and README.md quote for the headline kernel and the bench line equal what
the tracked files under profiles/ contain (tools/check_figures.py: every sentence must be found, every number must match
the source within the rounding the text shows)."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quoted_figures_match_the_tracked_profiles():
    spec = importlib.util.spec_from_file_location("check_figures", os.path.join(ROOT, "tools", "check_figures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    failures = mod.main(verbose=False)
    assert not failures, "\n".join(failures)


def test_a_wrong_figure_is_caught(tmp_path, monkeypatch):
    """The check really reads the documents: a copy of the repository's docs with one digit changed fails."""
    spec = importlib.util.spec_from_file_location("check_figures_t", os.path.join(ROOT, "tools", "check_figures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name in ("DESIGN.md", "README.md"):
        text = open(os.path.join(ROOT, name)).read()
        if name == "DESIGN.md":
            m = re.search(r"SIMD-cycles = (0\.\d{3})", text)
            assert m, "the sentence quoting the MFMA-busy counter is gone"
            wrong = f"{float(m.group(1)) + 0.01:.3f}"
            text = text.replace(m.group(0), f"SIMD-cycles = {wrong}")
        (tmp_path / name).write_text(text)
    os.symlink(os.path.join(ROOT, "profiles"), tmp_path / "profiles")
    monkeypatch.setattr(mod, "ROOT", str(tmp_path))
    failures = mod.main(verbose=False)
    assert len(failures) == 1 and wrong in failures[0]
