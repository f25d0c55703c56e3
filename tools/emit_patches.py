#!/usr/bin/env python3
"""The bindings of integration/mrbayes/ as UNIFIED DIFFS against a reference tree -- what a maintainer would review and apply
(`patch -p1 < mbamd_<binding>.diff` in the root of NBISweden/MrBayes).  The repository stores no reference text: the four source
transformers under integration/mrbayes/patches/ locate their edits by the reference's own function names and comments, and this
tool runs them on the tree it is given and writes what they changed.

    python tools/emit_patches.py [--reference /root/reference] <output directory>

tests/test_patches.py runs it in the CPU suite: every diff must apply to a pristine copy with `patch --dry-run`, give exactly the
transformer's output, and the reference files must be the ones the bindings were validated against (PINNED below).
"""
import difflib
import hashlib
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHES = os.path.join(ROOT, "integration", "mrbayes", "patches")

# binding -> [(reference file, transformer module, function)] in application order (a file two bindings edit: chained)
BINDINGS = {
    "pars": [("src/proposal.c", "patch_pars", "patch"), ("src/model.c", "patch_pars", "patch_model")],
    "reports": [("src/mcmc.c", "patch_reports", "patch_mcmc"), ("src/mbbeagle.c", "patch_reports", "patch_mbbeagle")],
    "eigen": [("src/likelihood.c", "patch_eigen", "patch")],
    "std": [("src/likelihood.c", "patch_std", "patch"), ("src/mcmc.c", "patch_std", "patch_mcmc")],
}
# sha256 of the reference files the bindings were validated against (NBISweden/MrBayes as surveyed: SURVEY.md).  An upstream edit
# to one of them changes its digest: the transformers may still apply, but the binding has to be looked at again.
PINNED = {
    "src/proposal.c": "4adb920603cb507ba7f04f6fa6e20e2bec80b8e7a602d5cdfb09f3b15902d18d",
    "src/model.c": "d980730737b85817ff96e4416ecb59ebb86a8a9b24f5c186437dca2b7956e2c3",
    "src/mcmc.c": "79f9831d55b84c57f97cd46608ca51f6d3f0d417e25fb3241291c54196dc83f5",
    "src/mbbeagle.c": "e7c9a934f047fe0d80276dc3aeb3194a61feaeaff2ae21b0f91b0c7de82f8706",
    "src/likelihood.c": "7af65d60c157fb5b7423b56c1c682dcfb3567192f0fee74ba0498d553c2997c5",
}


def _load(module):
    spec = importlib.util.spec_from_file_location(module, os.path.join(PATCHES, module + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.path.insert(0, PATCHES)
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(PATCHES)
    return mod


def digest(path):
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def binding_diffs(reference, binding):
    """[(relative file, original text, transformed text, unified diff text)] of one binding."""
    out = []
    for rel, module, fn in BINDINGS[binding]:
        with open(os.path.join(reference, rel)) as fh:
            before = fh.read()
        after = getattr(_load(module), fn)(before)
        diff = "".join(difflib.unified_diff(before.splitlines(True), after.splitlines(True), "a/" + rel, "b/" + rel, n=3))
        out.append((rel, before, after, diff))
    return out


def main():
    args = sys.argv[1:]
    reference = "/root/reference"
    if args[:1] == ["--reference"]:
        reference, args = args[1], args[2:]
    if len(args) != 1:
        raise SystemExit(__doc__)
    os.makedirs(args[0], exist_ok=True)
    for rel, want in PINNED.items():
        got = digest(os.path.join(reference, rel))
        if got != want:
            print("note: %s is not the file the bindings were validated against (sha256 %s, pinned %s)" % (rel, got[:16], want[:16]))
    for binding in BINDINGS:
        text = "".join(d for _, _, _, d in binding_diffs(reference, binding))
        path = os.path.join(args[0], "mbamd_%s.diff" % binding)
        with open(path, "w") as fh:
            fh.write(text)
        print("%s: %d hunks" % (path, text.count("\n@@ ")))


if __name__ == "__main__":
    main()
