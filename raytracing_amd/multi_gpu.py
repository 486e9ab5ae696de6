"""Frame partition over the GPUs of one node and the exchange that rebuilds the
picture.  GPU g of G renders the 8-row bands b with b % G == g, the
generalisation of the reference's `(thread_id..height).step_by(num_threads)`
row interleave (src/raytracer.rs:245); each rank quantises its own bands into a
compact block (rt_export_packed_bands) and ONE all-gather (RCCL over xGMI on
GPUs, gloo in the CPU tests) delivers all blocks to every rank, like the
channel fan-in of src/raytracer.rs:176-185.  No other collective touches the
data path."""
import torch
import torch.distributed as dist

BAND_ROWS = 8


def total_bands(height):
    return (height + BAND_ROWS - 1) // BAND_ROWS


def bands_of_rank(rank, world, height):
    return list(range(rank, total_bands(height), world))


def block_rows(world, height):
    """rows of the per-rank block: every rank pads to the band count of rank 0"""
    return len(bands_of_rank(0, world, height)) * BAND_ROWS


def display_row_index(world, height, device="cpu"):
    """index[r, m] = display row that row m of rank r's block holds, or `height`
    (one past the end) for padding rows.  Row m of a block is band r + (m//8)*world,
    line m%8 counted from the bottom of the picture."""
    rows = block_rows(world, height)
    m = torch.arange(rows, device=device)
    r = torch.arange(world, device=device).unsqueeze(1)
    j = (r + (m // BAND_ROWS).unsqueeze(0) * world) * BAND_ROWS + (m % BAND_ROWS).unsqueeze(0)
    return torch.where(j < height, height - 1 - j, torch.full_like(j, height))


def compose(gathered, height):
    """gathered: (world, block_rows, width) packed pixels -> (height, width) in display order"""
    world, rows, width = gathered.shape
    idx = display_row_index(world, height, gathered.device).reshape(-1)
    out = torch.zeros((height + 1, width), dtype=gathered.dtype, device=gathered.device)
    out.index_copy_(0, idx, gathered.reshape(world * rows, width))
    return out[:height]


def gather_frame(block, height, group=None):
    """all-gather the per-rank blocks and compose the full picture on every rank"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return compose(block.unsqueeze(0), height)
    rows, width = block.shape
    gathered = torch.empty((world * rows, width), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(gathered, block.contiguous(), group=group)   # rank r's block at rows r*rows..
    return compose(gathered.view(world, rows, width), height)
