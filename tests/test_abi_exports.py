"""The product library (mrbayes_amd/libhmsbeagle.so, built by hipcc for gfx950) loads without a GPU and exports every entry
point the headers under include/ declare -- the C ABI a libhmsbeagle client (MrBayes' src/mbbeagle.c) and the parsimony
binding link against.  No compute calls here: those need a device (-m gpu)."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for header in sorted(glob.glob(os.path.join(ROOT, "include", "**", "*.h"), recursive=True)):
        with open(header) as fh:
            text = fh.read()
        names += re.findall(r"BEAGLE_DLLEXPORT\s+[\w\s\*]+?\b(\w+)\s*\(", text)
    return names


def test_headers_declare_the_expected_surface():
    names = _declared()
    for must in ("beagleCreateInstance", "beagleUpdatePartials", "beagleCalculateEdgeLogLikelihoods", "beagleGetSiteLogLikelihoods",
                 "beagleUpdatePartialsByPartition", "beagleSetPatternPartitions", "mbamdSetRateMatrices",
                 "mbamdParsCreateInstance", "mbamdParsDownPass", "mbamdParsFinalPass", "mbamdParsScore",
                 "mbamdUpdateFinalPartials", "mbamdGetScaledPartials", "mbamdGetStepTiming"):
        assert must in names, must
    assert len(names) == len(set(names)) and len(names) >= 55


def test_product_library_exports_every_declared_symbol():
    from mrbayes_amd import build as mbbuild
    lib = ctypes.CDLL(mbbuild.build_library())
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    lib.beagleGetVersion.restype = ctypes.c_char_p
    assert b"mbamd" in lib.beagleGetVersion()


def test_no_device_means_an_error_not_a_fallback():
    """There is no CPU path in the product: without a HIP device the create calls fail (this test runs where no GPU is)."""
    from mrbayes_amd import build as mbbuild
    lib = ctypes.CDLL(mbbuild.build_library())
    n = ctypes.c_int(0)
    hip = ctypes.CDLL("libamdhip64.so")
    if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
        import pytest
        pytest.skip("a HIP device is present")
    lib.beagleCreateInstance.argtypes = [ctypes.c_int] * 9 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]
    assert lib.beagleCreateInstance(4, 8, 4, 4, 100, 1, 8, 4, 4, None, 0, 0, 0, None) < 0
    assert lib.mbamdParsCreateInstance(8, 100, 1, 4, -1) < 0
