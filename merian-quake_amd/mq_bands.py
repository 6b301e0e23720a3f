"""Row partition of the ReSTIR DI node and the post chain over the ranks of one node: who sends which rows to whom.

Plumbing around the C ABI (include/mq.h, "row partition"): libmqhip decides the bands (mq_band_layout) and owns every buffer
(mq_map_halo: full-size, row-indexed images on every rank); this module turns the bands into a list of point-to-point row copies
and issues them with torch.distributed -- `batch_isend_irecv`, i.e. grouped ncclSend / ncclRecv over RCCL: a band's halo rows
live on its neighbours, and xGMI gives every pair of GPUs its own link, so each copy runs on a link nobody else uses.
Per frame and inner rank at 3840x2160 on 8 ranks (spatial radius 30 = 32 rows, reprojection halo 64 rows: 96 rows on either
side of a band): reservoirs 2 x 96 rows x 3840 px x 64 B = 47.2 MB, accumulated image + history 2 x 96 x 3840 x 20 B = 14.7 MB
-- against 531 MB + 166 MB if the images themselves were gathered (bench.py --config5 uses a halo of H / 16 = 135 rows: its
fly-through moves pixels by up to 118 rows per frame; 108 MB).
"""
import numpy as np


def bands_of(ctx, W, H, world):
    """mq_band_layout for every rank of the partition: a list of mqhip.Band"""
    return [ctx.band_layout(W, H, r, world) for r in range(world)]


def _cut(a0, a1, b0, b1):
    lo, hi = max(a0, b0), min(a1, b1)
    return (lo, hi) if hi > lo else None


def plan(bands, me):
    """(sends, recvs) of rank `me`: lists of (peer, row_begin, row_end).  A rank needs the rows of [need_begin, need_end) it
    does not own, from whoever owns them; it sends the rows it owns to every rank that needs them."""
    sends, recvs = [], []
    for peer, pb in enumerate(bands):
        if peer == me:
            continue
        mine, theirs = bands[me], pb
        for lo, hi in ((mine.need_begin, mine.row_begin), (mine.row_end, mine.need_end)):  # what I need around my band ...
            c = _cut(lo, hi, theirs.row_begin, theirs.row_end)                                # ... of what the peer owns
            if c:
                recvs.append((peer, c[0], c[1]))
        for lo, hi in ((theirs.need_begin, theirs.row_begin), (theirs.row_end, theirs.need_end)):
            c = _cut(lo, hi, mine.row_begin, mine.row_end)
            if c:
                sends.append((peer, c[0], c[1]))
    return sends, recvs


class _DevRows:
    """Zero-copy [rows, row_bytes] uint8 view of a device image for torch.as_tensor (__cuda_array_interface__)."""

    def __init__(self, ptr, rows, row_bytes):
        self.__cuda_array_interface__ = {"shape": (rows, row_bytes), "typestr": "|u1", "data": (ptr, False), "version": 2}


def halo_tensors(ctx, H, which_list):
    """[(send image, recv image)] as [H, row_bytes] uint8 device tensors over libmqhip's own buffers (no copies)"""
    import torch
    out = []
    for which in which_list:
        send, recv, row_bytes = ctx.map_halo(which)
        out.append((torch.as_tensor(_DevRows(send, H, row_bytes), device="cuda"), torch.as_tensor(_DevRows(recv, H, row_bytes), device="cuda")))
    return out


def exchange(dist, pairs, sends, recvs, stage=None, group=None):
    """One halo exchange of this rank: for every (send image, recv image) pair move the planned rows.  `stage`: a function
    tensor -> tensor applied to what is sent and inverted on what is received (the gloo rehearsal stages through the host).
    `group`: a process group of its own keeps these transfers off the stream of the frame's all-gathers."""
    ops, landed = [], []
    for send_img, recv_img in pairs:
        for peer, r0, r1 in sends:
            t = send_img[r0:r1]
            ops.append(dist.P2POp(dist.isend, stage(t) if stage else t, peer, group))
        for peer, r0, r1 in recvs:
            t = recv_img[r0:r1]
            if stage:
                buf = stage(t)
                landed.append((t, buf))
                t = buf
            ops.append(dist.P2POp(dist.irecv, t, peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for dst, buf in landed:
        dst.copy_(buf)


def exchange_local(pairs_per_rank, bands):
    """The same exchange between K contexts that share ONE device (the tests' emulation of K ranks): plain row copies."""
    for me in range(len(bands)):
        _, recvs = plan(bands, me)
        for k, (_, recv_img) in enumerate(pairs_per_rank[me]):
            for peer, r0, r1 in recvs:
                recv_img[r0:r1].copy_(pairs_per_rank[peer][k][0][r0:r1])


_blocks = {}


def gather_rows(dist, image, bands, me, stage=None, group=None):
    """All ranks end up with every rank's owned rows of `image` ([H, row_bytes] tensor, rows a rank owns valid on that rank):
    one all_gather_into_tensor of equal-sized row blocks (the bands differ by at most one tile row: padded to the largest)."""
    import torch
    most = max(b.row_end - b.row_begin for b in bands)
    key = (most, image.shape[1], str(image.device), len(bands))
    if key not in _blocks:  # the staging blocks live as long as the process: no allocation, no fill per frame
        _blocks[key] = (torch.zeros((most, image.shape[1]), dtype=image.dtype, device=image.device), torch.empty((len(bands) * most, image.shape[1]), dtype=image.dtype, device=image.device))
    blk, allb_dev = _blocks[key]
    mine = bands[me]
    blk[: mine.row_end - mine.row_begin].copy_(image[mine.row_begin:mine.row_end])
    if stage:
        all_cpu = torch.empty((len(bands) * most, image.shape[1]), dtype=image.dtype)
        dist.all_gather_into_tensor(all_cpu, blk.cpu(), group=group)
        allb = all_cpu.to(image.device)
    else:
        allb = allb_dev
        dist.all_gather_into_tensor(allb, blk, group=group)
    for r, b in enumerate(bands):
        if r != me and b.row_end > b.row_begin:
            image[b.row_begin:b.row_end].copy_(allb[r * most: r * most + (b.row_end - b.row_begin)])
    return image


def halo_bytes(bands, me, row_bytes_list):
    """bytes this rank receives per frame"""
    _, recvs = plan(bands, me)
    return sum((r1 - r0) for _, r0, r1 in recvs) * sum(row_bytes_list)
