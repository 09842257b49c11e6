"""`dirt.projection` over torch tensors (dirt/projection.py:1-71): rays through pixels."""
import torch


def _pixel_to_ndc(pixel_locations, image_size):
    # pixel (0, 0) is the top-left corner of the image; NDC y points up (dirt/projection.py:6-7)
    flip = torch.tensor([1., -1.], dtype=pixel_locations.dtype, device=pixel_locations.device)
    return (-1. + 2. * pixel_locations / image_size) * flip


def _unproject_ndc_to_world(x_ndc, clip_to_world_matrix):
    # x_ndc [*, 3] (not homogeneous), row vectors (dirt/projection.py:10-20)
    x_h = torch.cat([x_ndc, torch.ones_like(x_ndc[..., :1])], dim=-1)
    x_world = torch.matmul(x_h[..., None, :], clip_to_world_matrix)[..., 0, :]
    return x_world[..., :3] / x_world[..., 3:]


def unproject_pixels_to_rays(pixel_locations, clip_to_world_matrix, image_size, name=None):
    """World-space start points (on the near plane) and unnormalised directions of the rays through the given
    pixel locations (dirt/projection.py:23-71).

    pixel_locations [A1..An, B1..Bm, 2] (x, y); clip_to_world_matrix [A1..An, 4, 4], typically
    inverse(world_to_view @ projection); image_size [A1..An, 2] = (width, height)."""
    pixel_locations = torch.as_tensor(pixel_locations, dtype=torch.float32)
    dev = pixel_locations.device
    clip_to_world_matrix = torch.as_tensor(clip_to_world_matrix, dtype=torch.float32, device=dev)
    image_size = torch.as_tensor(image_size, device=dev).to(torch.float32)
    per_iib_dims = pixel_locations.dim() - image_size.dim()  # m in the docstring
    image_size = image_size.reshape(image_size.shape[:-1] + (1,) * per_iib_dims + (2,))
    clip_to_world_matrix = clip_to_world_matrix.reshape(clip_to_world_matrix.shape[:-2] + (1,) * per_iib_dims + (4, 4))
    ndc_xy = _pixel_to_ndc(pixel_locations, image_size)
    starts = _unproject_ndc_to_world(torch.cat([ndc_xy, -torch.ones_like(ndc_xy[..., :1])], dim=-1), clip_to_world_matrix)
    deltas = _unproject_ndc_to_world(torch.cat([ndc_xy, torch.zeros_like(ndc_xy[..., :1])], dim=-1), clip_to_world_matrix) - starts
    return starts, deltas
