"""The bindings of integration/mrbayes/ as unified diffs (tools/emit_patches.py): each diff, made from the transformers' output on
the reference tree, applies to a pristine copy with `patch --dry-run`, a real `patch` gives exactly the transformer's text, and the
reference files are the ones the bindings were validated against -- an upstream edit fails HERE, with the file's name, not at run
time inside a regular expression.  Build container only (the reference tree is not on the GPU box)."""
import os
import shutil
import subprocess
import tempfile

import pytest

from tools import emit_patches

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference sources not present (build container only)")


def test_reference_files_are_the_pinned_ones():
    for rel, want in emit_patches.PINNED.items():
        assert emit_patches.digest(os.path.join(REF, rel)) == want, "%s changed upstream: validate the bindings against it again, then re-pin" % rel


@pytest.mark.parametrize("binding", sorted(emit_patches.BINDINGS))
def test_unified_diff_applies(binding):
    if shutil.which("patch") is None:
        pytest.skip("no patch(1)")
    for rel, before, after, diff in emit_patches.binding_diffs(REF, binding):
        assert diff.count("\n@@ ") + diff.startswith("@@ ") >= 1 and before != after, rel
        with tempfile.TemporaryDirectory() as wd:
            os.makedirs(os.path.join(wd, os.path.dirname(rel)))
            target = os.path.join(wd, rel)
            with open(target, "w") as fh:
                fh.write(before)
            with open(os.path.join(wd, "b.diff"), "w") as fh:
                fh.write(diff)
            dry = subprocess.run(["patch", "-p1", "--dry-run", "-i", "b.diff"], cwd=wd, capture_output=True, text=True)
            assert dry.returncode == 0 and "FAILED" not in dry.stdout and "fuzz" not in dry.stdout, dry.stdout + dry.stderr
            with open(target) as fh:
                assert fh.read() == before                   # (a dry run)
            real = subprocess.run(["patch", "-p1", "-i", "b.diff"], cwd=wd, capture_output=True, text=True)
            assert real.returncode == 0, real.stdout + real.stderr
            with open(target) as fh:
                assert fh.read() == after, rel
