"""Slice-per-GPU sharding of a picture and the one exchange step it needs.

JM shards a picture only by slice: `SliceMode=1`, `SliceArgument = k * PicWidthInMbs` gives contiguous bands of k
macroblock rows (lencod/src/slice.c:431 terminates a slice after SliceArgument macroblocks; configs[3] of BASELINE.json
uses 8 bands of a 2160p picture).  Within a P picture the bands are independent for motion estimation, transform /
quantisation and -- with DFDisableIdc=2 (lencod/src/loopFilter.c:159-165) -- deblocking; the only coupling is that
the NEXT picture's motion search reads the whole reconstructed reference.  So each rank keeps its own band, and
after deblocking every rank all-gathers the reconstructed bands (RCCL over xGMI on the GPU box, gloo in the CPU
tests) and cuts out the rows its own search windows can reach: its band plus a halo above and below.

Nothing in here computes samples; it is index arithmetic plus one collective.
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


def halo_rows(search_range, max_pred_pels):
    """Rows above / below a band that its search windows can read: the search centre may sit `max_pred_pels` away from the
    block (the level's vertical MV limit, lencod/src/conformance.c:604-631), the window adds `search_range`, a block 16, and
    the 6-tap interpolation 3; rounded up to whole macroblock rows."""
    return -(-(max_pred_pels + search_range + 16 + 3) // 16) * 16


def reference_rows(band, halo, pic_height):
    """Row indices of the full picture that make up the band's local reference: [y0 - halo, y0 + height + halo) clamped
    into the picture (rows outside the picture are edge replicas, exactly what JM's padded planes hold there)."""
    idx = torch.arange(band.y0 - halo, band.y0 + band.height + halo)
    return idx.clamp_(0, pic_height - 1)


def exchange_reference(own_band_rows, band, halo, pic_height, group=None, gathered=None, out=None, idx=None):
    """All-gather the reconstructed bands and return this rank's local reference (band + halos).

    own_band_rows : (16 * rows_per_band, W) uint8 -- the rank's reconstructed band, padded with arbitrary rows when the
                    band is shorter than rows_per_band (all_gather needs equal shapes)
    returns       : (band.height + 2 * halo, W) uint8
    The one collective of the path; `gathered` / `out` / `idx` (= reference_rows(...) already on the device) let the caller reuse
    buffers across pictures, so that a step issues exactly one collective and one gather kernel and no host-to-device copy."""
    world = dist.get_world_size(group)
    rows, width = own_band_rows.shape
    assert rows == 16 * band.rows_per_band, (rows, band)
    if gathered is None:
        gathered = torch.empty((world, rows, width), dtype=own_band_rows.dtype, device=own_band_rows.device)
    dist.all_gather_into_tensor(gathered.view(-1), own_band_rows.contiguous().view(-1), group=group)
    # bands are laid out back to back with a stride of 16*rows_per_band rows; picture row y lives at the same index as long as
    # every band but the last is full, which is how band_of() partitions
    tall = gathered.view(world * rows, width)
    if idx is None:
        idx = reference_rows(band, halo, pic_height).to(own_band_rows.device)
    if out is None:
        out = torch.empty((idx.numel(), width), dtype=own_band_rows.dtype, device=own_band_rows.device)
    torch.index_select(tall, 0, idx, out=out)
    return out


def packed_band(luma_rows, u_rows, v_rows):
    """One band's reconstruction as ONE (rows + rows/2, W) uint8 tensor for the exchange: the luma rows, then the 4:2:0 chroma rows with
    U in the left half and V in the right half of each row (16 * rows_per_band luma rows -> 8 * rows_per_band chroma rows)."""
    return torch.cat([luma_rows, torch.cat([u_rows, v_rows], dim=1)], dim=0)


class YuvExchange:
    """The one collective of the path for 4:2:0 pictures: all-gather of the packed bands (luma + chroma: 1.5 bytes per sample position),
    then one gather kernel per plane kind cutting out this rank's local reference: luma rows [y0 - halo, y0 + height + halo) and chroma
    rows [(y0 - halo) / 2, (y0 + height + halo) / 2), clamped into the picture.  Buffers and index tensors are built once."""

    def __init__(self, band, halo, pic_height, width, world, device):
        self.band, self.world, self.width = band, world, width
        self.rows = 16 * band.rows_per_band                  # luma rows per (padded) band
        self.per_rank = self.rows + self.rows // 2           # rows of one packed band
        y = reference_rows(band, halo, pic_height)           # picture rows of the local luma reference
        c = torch.arange((band.y0 - halo) // 2, (band.y0 + band.height + halo) // 2).clamp_(0, pic_height // 2 - 1)
        # picture row -> (rank that owns it, row inside that rank's packed band); every band but the last is full
        self.idx_y = ((y // self.rows) * self.per_rank + y % self.rows).to(device)
        self.idx_c = ((c // (self.rows // 2)) * self.per_rank + self.rows + c % (self.rows // 2)).to(device)
        self.gathered = torch.empty((world * self.per_rank, width), dtype=torch.uint8, device=device)
        self.out_y = torch.empty((len(y), width), dtype=torch.uint8, device=device)
        self.out_c = torch.empty((len(c), width), dtype=torch.uint8, device=device)

    def __call__(self, packed, group=None):
        """packed: packed_band(...) of this rank, (per_rank, W); returns (local luma reference, local U, local V)"""
        assert packed.shape == (self.per_rank, self.width), (packed.shape, self.per_rank, self.width)
        dist.all_gather_into_tensor(self.gathered.view(-1), packed.contiguous().view(-1), group=group)
        torch.index_select(self.gathered, 0, self.idx_y, out=self.out_y)
        torch.index_select(self.gathered, 0, self.idx_c, out=self.out_c)
        return self.out_y, self.out_c[:, : self.width // 2], self.out_c[:, self.width // 2:]


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


class PictureGather(BandGather):
    """BandGather of a 4:2:0 picture's three planes: rank r owns luma rows [r * band_rows, (r + 1) * band_rows) and chroma rows
    [r * band_rows / 2, ...); band_rows defaults to equal bands."""

    def __init__(self, y, u, v, world, rank, group=None, band_rows=None):
        if band_rows is None:
            assert y.shape[0] % world == 0 and u.shape[0] % world == 0, (y.shape, u.shape, world)
            band_rows = y.shape[0] // world
        assert v.shape == u.shape and band_rows % 2 == 0
        super().__init__([(y, band_rows), (u, band_rows // 2), (v, band_rows // 2)], world, rank, group)
