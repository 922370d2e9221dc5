"""Several GPUs, one process each.  Two ways to use them, both index arithmetic plus at most one collective:

* **Closed GOPs** (`gop_of`): with IDRPeriod = K the pictures of GOP g -- an IDR picture and its K - 1 P pictures -- refer to nothing outside it (lencod/src/image.c:
  an IDR empties the DPB, idr_memory_management mbuffer.c), so rank r codes GOPs r, r + N, ... of the sequence with NO collective on the data path; the bitstream is the GOPs'
  NAL units in order, which is what JM writes with the same IDRPeriod.  `bench.py --gpus N` measures this (weak scaling: one GOP per GPU per round).
* **Slices of one picture** (`band_of`, `BandGather`): JM shards a picture only by slice: `SliceMode = 1`, `SliceArgument = k * PicWidthInMbs` gives contiguous bands of k
  macroblock rows (lencod/src/slice.c:431; BASELINE configs[3]: 8 bands of a 2160p picture).  The bands are coded independently; DeblockFrame filters across slice edges
  (DFDisableIdc = 0 in every shipped .cfg: lencod/src/loopFilter.c:159-165) and the next picture's search windows reach into the neighbouring bands, so after coding every
  rank all-gathers the UN-deblocked bands and their loop-filter side information (`BandGather`: RCCL over xGMI on the GPU box, gloo in the CPU tests), deblocks the whole
  picture and keeps the whole reference.  In ONE process with a context per device the same exchange is `jmhip_allgather_bands` (include/jmhip.h; peer copies).
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Band:
    rank: int
    first_mb_row: int      # first macroblock row of the band in the picture
    mb_rows: int           # macroblock rows in the band (the last band may be shorter)
    rows_per_band: int     # k = ceil(total / n): SliceArgument / PicWidthInMbs

    @property
    def y0(self):
        return 16 * self.first_mb_row

    @property
    def height(self):
        return 16 * self.mb_rows


def slice_argument(pic_height_in_mbs, pic_width_in_mbs, n_bands):
    """JM's SliceArgument for n_bands equal bands: ceil(rows / n) * PicWidthInMbs macroblocks per slice."""
    return -(-pic_height_in_mbs // n_bands) * pic_width_in_mbs


def band_of(rank, n_bands, pic_height_in_mbs):
    """The band JM's slice partitioning gives rank `rank` of `n_bands`."""
    k = -(-pic_height_in_mbs // n_bands)
    first = min(rank * k, pic_height_in_mbs)
    return Band(rank, first, max(0, min(k, pic_height_in_mbs - first)), k)


def gop_of(rank, world, n_pictures, idr_period):
    """The pictures rank `rank` of `world` codes when a sequence of n_pictures is cut at its IDR pictures (IDRPeriod = idr_period): [(first picture, pictures)] of GOPs
    rank, rank + world, ...; every picture belongs to exactly one rank, a GOP is never split."""
    assert idr_period > 0 and world > 0 and 0 <= rank < world
    n_gops = -(-n_pictures // idr_period)
    return [(g * idr_period, min(idr_period, n_pictures - g * idr_period)) for g in range(rank, n_gops, world)]


class BandGather:
    """Every rank ends up with every band: one all-gather of bands of rows of several row-major planes, the last band possibly shorter
    (JM's SliceArgument partitioning: 2160p in 8 slices of 4080 macroblocks is 7 bands of 17 macroblock rows and one of 16).

    planes: [(tensor (rows, pitch) uint8, rows_per_band)], rank r owns rows [r * k, min((r + 1) * k, rows)) of each.  The tensors may alias library
    memory (the pipeline's reconstruction and loop filter side information).  pack -> all_gather_into_tensor -> strided copies back; the packed
    buffer keeps the collective's input and output from aliasing and pads the short band, so a picture costs exactly one collective."""

    def __init__(self, planes, world, rank, group=None):
        self.planes, self.world, self.rank, self.group = planes, world, rank, group
        self.off, n = [], 0
        for t, k in planes:
            assert t.dim() == 2 and t.dtype == torch.uint8 and k > 0 and (world - 1) * k < t.shape[0] <= world * k, (t.shape, k, world)
            self.off.append(n)
            n += k * t.shape[1]
        dev = planes[0][0].device
        self.own = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.all = torch.empty((world, n), dtype=torch.uint8, device=dev)

    def __call__(self):
        r = self.rank
        for (t, k), o in zip(self.planes, self.off):
            mine = t[r * k:(r + 1) * k]
            self.own[o:o + mine.numel()].view(mine.shape).copy_(mine)
        dist.all_gather_into_tensor(self.all.view(-1), self.own, group=self.group)
        for (t, k), o in zip(self.planes, self.off):
            rows, pitch = t.shape
            full = rows // k
            t[:full * k].view(full, k * pitch).copy_(self.all[:full, o:o + k * pitch])
            if rows > full * k:
                t[full * k:].copy_(self.all[full, o:o + (rows - full * k) * pitch].view(rows - full * k, pitch))
