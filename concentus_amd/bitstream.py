"""The reference CLI's packet container (`opus_demo` .bit files), so batches round-trip through the reference's own
tools: per packet `[be32 length][be32 encoder final range][payload]` (writer opus-fix/src/opus_demo.c:747-765,
reader :653-672). Host-side framing only; the payload bytes are what opusgpu_encode_batch produced."""
import struct

import numpy as np


def write_opus_demo_bit(path, packets, lengths, final_ranges):
    """packets: uint8 [n][stride] (host array), lengths int [n], final_ranges uint32 [n]: one stream, in order."""
    packets = np.asarray(packets)
    with open(path, "wb") as f:
        for k in range(len(lengths)):
            n = int(lengths[k])
            if n < 0:
                raise ValueError("packet %d carries error %d" % (k, n))
            f.write(struct.pack(">iI", n, int(final_ranges[k]) & 0xFFFFFFFF))
            f.write(packets[k, :n].tobytes())


def read_opus_demo_bit(path, stride=1276):
    """Returns (packets uint8 [n][stride], lengths int32 [n], final_ranges uint32 [n])."""
    pk, ln, rg = [], [], []
    with open(path, "rb") as f:
        while True:
            h = f.read(8)
            if len(h) < 8:
                break
            n, r = struct.unpack(">iI", h)
            if n < 0 or n > stride:
                raise ValueError("invalid payload length %d" % n)
            d = f.read(n)
            if len(d) < n:
                raise ValueError("ran out of input")
            row = np.zeros(stride, np.uint8)
            row[:n] = np.frombuffer(d, np.uint8)
            pk.append(row)
            ln.append(n)
            rg.append(r)
    return (np.stack(pk) if pk else np.zeros((0, stride), np.uint8)), np.array(ln, np.int32), np.array(rg, np.uint32)
