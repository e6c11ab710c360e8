"""The device hand-off kernel (omniparser_amd/csrc/glue_ops.hip::glue_kernel) run FROM ITS DEVICE SOURCE on the host emulation
(tests/emu): the GPU check `gpu_checks.check_glue` — 36 fixtures recorded from the reference's own remove_overlap_new /
int_box_area, 40 random trials against the host twin ScreenParser.glue — runs against it unchanged.  The `-m gpu` twin of this test
launches the same source on the MI355X."""
import pytest


def test_device_handoff_source_on_host_matches_reference_fixtures_and_host_twin(emu):
    import gpu_checks as G
    r = G.check_glue()
    assert r["fixture_cases"] >= 36 and r["random_trials"] >= 40


def test_emulated_kernel_rejects_bad_capacity(emu):
    from omniparser_amd import _lib as L
    op = L.make_op(L.OP_GLUE, L.F32, p=[0] * 8, i={0: 300, 1: 10, 2: 1920, 3: 1080, 4: 1, 6: 310})
    with pytest.raises(L.OmniError, match="null pointer"):
        L.launch(op)
