"""Helpers of the GPU parity tests: twin-stream comparison with a diagnosis.

Size-independent property of the encode path (tests/test_full_size_gpu.py, tests/test_ordering_gpu.py): K distinct
signals are dealt round-robin over S streams, so stream s must produce, packet number by packet number, exactly the
packets of stream s % K — whatever lane, tile, batch, round or HIP stream its blocks went through.  The first K
streams are compared with the oracle by the caller."""
import numpy as np
import torch

_W = None


def _weights(n, device):
    global _W
    if _W is None or _W.shape[0] < n or _W.device != device:
        g = torch.Generator().manual_seed(12345)
        _W = (torch.randint(-(2 ** 62), 2 ** 62, (max(n, 1024),), generator=g, dtype=torch.int64) | 1).to(device)
    return _W[:n]


class TwinLedger:
    """Collects every packet that comes out (any order, any grouping) and checks the twin property.

    add(): rows whose twin's packet of the same packet number is in the same set are compared byte for byte at once,
    with a diagnosis (first differing stream, block type, byte offset, whether the offset lies inside the packet);
    every row also leaves a 64-bit hash, and finish() compares the whole per-stream sequences (a stream may deliver a
    block in a later round than its twin)."""

    def __init__(self, S, K, device):
        self.S, self.K, self.dev = S, K, device
        self.rec = []                       # (stream, packetno, mode, nbytes, hash) per add(), device tensors
        self.first = [dict() for _ in range(K)]     # packetno -> bytes, streams 0..K-1
        self.nblocks = 0
        self.modes = np.zeros(4, np.int64)

    def add(self, stream, packetno, mode, packets, nbytes, label=""):
        """stream / packetno / mode: integer tensors or arrays [n]; packets uint8 [n, maxb]; nbytes int32 [n]"""
        dev = self.dev
        stream = torch.as_tensor(np.ascontiguousarray(stream) if isinstance(stream, np.ndarray) else stream).to(dev).long()
        packetno = torch.as_tensor(np.ascontiguousarray(packetno) if isinstance(packetno, np.ndarray) else packetno).to(dev).long()
        mode = torch.as_tensor(np.ascontiguousarray(mode) if isinstance(mode, np.ndarray) else mode).to(dev).long()
        n = int(stream.shape[0])
        if n == 0:
            return
        nbytes = nbytes.to(dev)
        assert packets.shape[0] == n and nbytes.shape[0] == n
        assert bool((nbytes >= 0).all()), f"{label}: packet buffer overflow (length -1)"
        self.nblocks += n
        self.modes += np.bincount(mode.cpu().numpy(), minlength=4)[:4]
        maxb = packets.shape[1]
        # ---- twins inside this set
        key = stream * (1 << 24) + packetno
        tkey = (stream % self.K) * (1 << 24) + packetno
        skey, order = torch.sort(key)
        at = torch.searchsorted(skey, tkey).clamp(max=n - 1)
        found = skey[at] == tkey
        twin = order[at]
        rows = torch.nonzero(found & (twin != torch.arange(n, device=dev))).flatten()
        if rows.numel():
            a, b = packets[rows], packets[twin[rows]]
            same_len = nbytes[rows] == nbytes[twin[rows]]
            eq = (a == b)
            bad = ~(eq.all(dim=1) & same_len)
            if bool(bad.any()):
                k = int(torch.nonzero(bad).flatten()[0])
                i, j = int(rows[k]), int(twin[rows[k]])
                neq = ~eq[k]
                off = int(torch.nonzero(neq).flatten()[0]) if bool(neq.any()) else -1
                nb_i, nb_j = int(nbytes[i]), int(nbytes[j])
                badrows = rows[bad]
                where = "length differs" if nb_i != nb_j else ("INSIDE the packet" if off < nb_i else "BEYOND the packet's length (tail not zero)")
                raise AssertionError(
                    f"{label}: identical input, different packets: {int(bad.sum())} of {rows.numel()} rows differ; first: row {i} "
                    f"(lane {i % 64} of tile {i // 64}) stream {int(stream[i])} vs twin row {j} stream {int(stream[j])}, "
                    f"packetno {int(packetno[i])}, block type {int(mode[i])}, lengths {nb_i} / {nb_j}, first differing byte {off} "
                    f"of {maxb} ({where}), {int(neq.sum())} bytes of the row differ; "
                    f"differing rows (row, stream, type): {[(int(r), int(stream[r]), int(mode[r])) for r in badrows[:12]]}")
        # ---- hashes for the sequence check
        w = _weights(maxb // 8, dev)
        h = (packets.contiguous().view(torch.int64) * w).sum(dim=1) + nbytes.long() * 1000003
        self.rec.append((stream, packetno, mode, nbytes.long(), h))
        # ---- the reference streams' packets, for the oracle
        lead = torch.nonzero(stream < self.K).flatten()
        if lead.numel():
            pk = packets[lead].cpu().numpy()
            for r, s, pn, nb in zip(range(lead.numel()), stream[lead].tolist(), packetno[lead].tolist(), nbytes[lead].tolist()):
                assert pn not in self.first[s], f"{label}: stream {s} delivered packet {pn} twice"
                self.first[s][pn] = bytes(pk[r, :nb])

    def add_host_round(self, info, packets, nbytes, label=""):
        """outputs of FrontEnd.encode_round / encode_rounds (info: numpy records)"""
        self.add(info["stream"], info["packetno"], info["block_mode"], packets, nbytes, label)

    def add_device_rounds(self, info, packets, nbytes, label=""):
        """outputs of FrontEnd.encode_rounds_device (info: uint8 [lanes, 40] records on the device; -2 = empty lane)"""
        live = torch.nonzero(nbytes != -2).flatten()
        if live.numel() == 0:
            return 0
        rec = info.view(torch.int32)[live]                    # stream, block_mode, lW, W, nW, eos, granulepos(2), packetno(2)
        pno = info.view(torch.int64)[live][:, 4]
        self.add(rec[:, 0], pno, rec[:, 1], packets[live], nbytes[live], label)
        return int(live.numel())

    def finish(self, label=""):
        """every stream's sequence of (packetno, length, hash) equals its twin's"""
        if not self.rec:
            return
        stream = torch.cat([r[0] for r in self.rec]).cpu().numpy()
        pno = torch.cat([r[1] for r in self.rec]).cpu().numpy()
        mode = torch.cat([r[2] for r in self.rec]).cpu().numpy()
        nb = torch.cat([r[3] for r in self.rec]).cpu().numpy()
        h = torch.cat([r[4] for r in self.rec]).cpu().numpy()
        order = np.lexsort((pno, stream))
        stream, pno, mode, nb, h = stream[order], pno[order], mode[order], nb[order], h[order]
        assert not np.any((stream[1:] == stream[:-1]) & (pno[1:] == pno[:-1])), f"{label}: a packet number came twice"
        start = np.searchsorted(stream, np.arange(self.S + 1))
        cnt = np.diff(start)
        tw = np.arange(self.S) % self.K
        if not np.array_equal(cnt, cnt[tw]):
            s = int(np.flatnonzero(cnt != cnt[tw])[0])
            raise AssertionError(f"{label}: stream {s} delivered {cnt[s]} packets, its twin {tw[s]} delivered {cnt[tw[s]]}")
        # twin's entry of the same rank
        rank = np.arange(len(stream)) - start[stream]
        tidx = start[tw[stream]] + rank
        bad = (pno != pno[tidx]) | (nb != nb[tidx]) | (h != h[tidx])
        if bad.any():
            i = int(np.flatnonzero(bad)[0])
            raise AssertionError(
                f"{label}: identical input, different packets (sequence check): {int(bad.sum())} packets differ; first: stream "
                f"{stream[i]} packetno {pno[i]} block type {mode[i]} length {nb[i]} vs twin {tw[stream[i]]}: packetno "
                f"{pno[tidx[i]]} length {nb[tidx[i]]}; streams affected: {sorted(set(stream[bad].tolist()))[:16]}")

    def lead_packets(self, k):
        """packets of reference stream k in packet-number order"""
        return [self.first[k][pn] for pn in sorted(self.first[k])]
