"""__graft_entry__.smoke() as a test: the driver runs it on the MI355X before the bench; here it is part of the -m gpu suite (and
of its rehearsal on the functional model)."""
import pytest

pytestmark = pytest.mark.gpu


def test_smoke_entry_point(hiplib, cuda_device, capsys):
    import __graft_entry__ as g

    g.smoke()
    assert "smoke ok on gfx950" in capsys.readouterr().out
