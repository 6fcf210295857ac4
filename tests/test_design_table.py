"""The measurement table of DESIGN.md is the one scripts/design_table.py generates from the recorded bench lines."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_table_matches_the_recorded_bench_lines():
    spec = importlib.util.spec_from_file_location('design_table', os.path.join(ROOT, 'scripts', 'design_table.py'))
    dt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dt)
    text = open(os.path.join(ROOT, 'DESIGN.md')).read()
    assert dt.BEGIN.search(text), 'DESIGN.md has no generated table'
    assert dt.rewrite(text) == text, 'run: python scripts/design_table.py x --write'
