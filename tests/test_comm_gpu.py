"""GPU suite: the raw RCCL communicator behind the C ABI (pm_comm_*, prime_rl_amd/comm.py) on one MI355X: TCP-store bootstrap,
unique-id exchange, communicator creation and every collective the path uses, on a world of one rank (the collectives then copy, but
the whole call path — dlopen'ed librccl, ncclCommInitRank, stream-ordered enqueue — is the multi-GPU one).  Multi-rank behaviour of the
protocols built on it: tests/test_weight_push_cpu.py (gloo, 4 ranks, same communicator surface)."""

import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _port():
    from tests._ports import free_port

    return free_port()


def test_rccl_communicator_world1_all_ops():
    from prime_rl_amd import _lib
    from prime_rl_amd.comm import RCCLCommunicator, StatelessRendezvous

    lib = _lib.load()
    assert lib.pm_comm_available() == 1 and lib.pm_comm_rccl_version() > 20000
    comm = RCCLCommunicator(StatelessRendezvous("127.0.0.1", _port(), 0, 1, timeout=60), DEV)
    assert (comm.rank, comm.size) == (0, 1)
    side = torch.cuda.Stream(device=DEV)
    x = torch.arange(4096, dtype=torch.float32, device=DEV)
    with torch.cuda.stream(side):  # operations follow the CURRENT stream
        y = x.clone()
        comm.all_reduce(y)
        b = torch.arange(1000, dtype=torch.bfloat16, device=DEV)
        b0 = b.clone()
        comm.broadcast(b, 0)
        out = torch.empty(4096, dtype=torch.float32, device=DEV)
        comm.reduce_scatter(out, x)
        ag = torch.empty(4096, dtype=torch.float32, device=DEV)
        comm.all_gather(ag, x)
        a2a = torch.empty(4096, dtype=torch.float32, device=DEV)
        comm.all_to_all(a2a, x)
        i64 = torch.tensor([3, 5, 7], dtype=torch.int64, device=DEV)
        comm.all_reduce(i64)
        u8 = torch.arange(200, dtype=torch.uint8, device=DEV)
        r8 = torch.zeros_like(u8)
        comm.group_start()
        comm.send(u8, 0)
        comm.recv(r8, 0)
        comm.group_end()
    side.synchronize()
    for t in (y, out, ag, a2a):
        assert torch.equal(t, x)
    assert torch.equal(b, b0) and i64.tolist() == [3, 5, 7] and torch.equal(r8, u8)
    comm.barrier()
    with pytest.raises(_lib.PrimeMI355XError):
        comm.all_reduce(torch.zeros(4))  # host tensors are refused, nothing falls back
    comm.close()


def test_reference_protocol_over_rccl_world1():
    """The reference's byte protocol (count, pickled metadata, one flat tensor per dtype) through real RCCL broadcasts."""
    from prime_rl_amd.comm import RCCLCommunicator, StatelessRendezvous
    from prime_rl_amd.transports.weights import send_reference_protocol

    comm = RCCLCommunicator(StatelessRendezvous("127.0.0.1", _port(), 0, 1, timeout=60), DEV, name="ref")
    sd = {"lm_head.weight": torch.randn(8, 4, device=DEV).to(torch.bfloat16), "model.layers.0.w": torch.randn(4, 4, device=DEV).to(torch.bfloat16)}
    send_reference_protocol(comm, sd, DEV)  # world of one: every broadcast is a self-copy; the call path is what is exercised
    torch.cuda.synchronize()
    comm.close()
