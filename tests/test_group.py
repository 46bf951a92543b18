"""CPU: host logic of the single-process device group (gpx_group.inc / LocalHub in
gpx_shard.inc): the two-barrier rendezvous the rank threads of the LOCAL transport use —
nobody reads before everybody has published, nobody overwrites before everybody has read, and
an aborting rank wakes the others instead of deadlocking them.  No GPU involved."""
import ctypes as C

import pytest


@pytest.mark.parametrize("P,rounds", [(1, 10), (2, 2000), (4, 1000), (8, 1000)])
def test_hub_rounds_complete_and_see_consistent_data(gpx, P, rounds):
    done = C.c_int32(-1)
    assert gpx.gpx_debug_local_hub(P, rounds, -1, 0, C.byref(done)) == 0
    assert done.value == rounds


@pytest.mark.parametrize("P,who,when", [(2, 1, 0), (4, 0, 17), (8, 3, 77), (8, 7, 499)])
def test_hub_abort_releases_every_waiter(gpx, P, who, when):
    """Rank `who` fails instead of arriving at round `when`: every other rank returns (the call
    comes back at all), having completed exactly the rounds before it."""
    done = C.c_int32(-1)
    assert gpx.gpx_debug_local_hub(P, 500, who, when, C.byref(done)) == 0
    assert done.value == when


def test_hub_bad_arguments(gpx):
    done = C.c_int32(0)
    assert gpx.gpx_debug_local_hub(0, 1, -1, 0, C.byref(done)) == -1
    assert gpx.gpx_debug_local_hub(2, 1, -1, 0, None) == -1


def test_group_argument_validation():
    from gaussianprocesspathmodelling_amd import GP
    with pytest.raises(ValueError):
        GP(devices=0)
    with pytest.raises(ValueError):
        GP(devices=[])
    with pytest.raises(ValueError):
        GP(devices=list(range(9)))
    with pytest.raises(ValueError):
        GP(devices=2, transport="mpi")
    with pytest.raises(ValueError):
        GP(devices=2, world=2, rank=0)
