"""Host-side description of the multi-GPU framebuffer sharding, mirrored from
mq_api.cpp / mq_kernels.hip so the layout can be reasoned about (and tested) without a GPU.

The framebuffer is cut into 8x8-pixel tiles, numbered row-major.  Rank r of `world` renders the
tiles t with t % world == r; its exchange buffer MQ_OUT_TILES holds them in order of local tile
index lt = t // world, 64 RGBA32F pixels per tile (pixel (ix, iy) of the tile at ix + 8*iy).
Every rank's buffer is padded to tiles_per_rank = ceil(n_tiles / world) tiles so one
all_gather_into_tensor of equal-sized buffers moves the whole frame."""
import numpy as np

TILE = 8


def grid(W, H):
    return (W + TILE - 1) // TILE, (H + TILE - 1) // TILE


def tiles_per_rank(W, H, world):
    tx, ty = grid(W, H)
    return (tx * ty + world - 1) // world


def local_tiles(W, H, rank, world):
    tx, ty = grid(W, H)
    return np.arange(rank, tx * ty, world)


def tile_image(image, rank, world):
    """Pack the tiles of `rank` out of a full (H, W, 4) image into its exchange buffer."""
    H, W = image.shape[:2]
    tx, _ = grid(W, H)
    buf = np.zeros((tiles_per_rank(W, H, world), TILE * TILE, 4), image.dtype)
    for lt, t in enumerate(local_tiles(W, H, rank, world)):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        blk = np.zeros((TILE, TILE, 4), image.dtype)
        sub = image[y0:y0 + TILE, x0:x0 + TILE]
        blk[:sub.shape[0], :sub.shape[1]] = sub
        buf[lt] = blk.reshape(TILE * TILE, 4)
    return buf


def untile(gathered, W, H, world):
    """Inverse of the gather: (world, tiles_per_rank, 64, 4) -> (H, W, 4)."""
    tx, ty = grid(W, H)
    gathered = np.asarray(gathered).reshape(world, -1, TILE * TILE, 4)
    out = np.zeros((H, W, 4), gathered.dtype)
    for t in range(tx * ty):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        blk = gathered[t % world, t // world].reshape(TILE, TILE, 4)
        h, w = min(TILE, H - y0), min(TILE, W - x0)
        out[y0:y0 + h, x0:x0 + w] = blk[:h, :w]
    return out
