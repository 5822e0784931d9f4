"""A TCP port for a test's rendezvous.  NOT `bind(("", 0))`: that hands out a port of the kernel's ephemeral range (32768-60999 here), and between closing
the probe socket and the TCPStore's listen() any outgoing connection on the box (gloo pairs of a neighbouring test, a lingering TIME_WAIT) may be given
the same number — `EADDRINUSE`, once per few hundred runs.  Ports below the ephemeral range are only ever taken by listeners."""
import os
import random
import socket


def free_port() -> int:
    rng = random.Random(os.getpid() * 7919 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        port = rng.randrange(20000, 32000)
        with socket.socket() as s:
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
        return port
    raise RuntimeError("no free port between 20000 and 32000")
