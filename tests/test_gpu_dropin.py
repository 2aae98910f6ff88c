"""GPU parity, file to file: the HIP drop-in executables against the reference goldens (and the oracle)."""
import os
import subprocess
import tempfile

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", gu.case_names())
def test_dropin_binaries_match_reference_goldens(built, case):
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        outs = gu.run_stage_pair([built["cv"]], [built["sr"]], td, meta)
        assert gu.compare(td, outs) == []


def test_native_library_is_what_ran(built):
    """The executables are linked against the in-tree HIP library (no site-packages copy, no fallback)."""
    out = subprocess.run(["ldd", built["cv"]], stdout=subprocess.PIPE).stdout.decode()
    assert "libhairsplitter_hip.so" in out and "hairsplitter_amd/lib" in out.replace("bin/../lib", "lib")
