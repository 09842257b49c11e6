"""`dirt.matrices` over torch tensors (dirt/matrices.py:1-208): homogeneous transform helpers.

Conventions are the reference's (dirt/matrices.py:3-8): matrices RIGHT-multiply row vectors, i.e. they are
indexed [*, in, out]; `*` is any number of leading batch dimensions.  This is small dense math upstream of the
rasteriser; it runs wherever its inputs live (the reference computes it in the TF graph) and is differentiable
through torch autograd."""
import torch


def _t(x, like=None):
    if isinstance(x, torch.Tensor):
        return x if x.is_floating_point() else x.to(torch.float32)
    return torch.as_tensor(x, dtype=torch.float32, device=like.device if isinstance(like, torch.Tensor) else None)


def pad_3x3_to_4x4(matrix, name=None):
    """[*, 3, 3] -> [*, 4, 4]: a zero column, a zero row, and a one in the corner (dirt/matrices.py:155-180)."""
    matrix = _t(matrix)
    out = matrix.new_zeros(matrix.shape[:-2] + (4, 4))
    out[..., :3, :3] = matrix
    out[..., 3, 3] = 1.
    return out


def rodrigues(vectors, name=None, three_by_three=False):
    """Angle-axis rotation matrices, [*, 3] -> [*, 4, 4] (or [*, 3, 3]) (dirt/matrices.py:15-61).

    Follows the reference exactly, including the 1e-12 offset it adds to the vector so that the derivative
    exists at zero, and its index convention R[in, out] = c*I + (1-c)*k k^T + s*K with K[in, out] the
    cross-product matrix laid out as in OpenCV's documentation."""
    v = _t(vectors) + 1.e-12
    norms = torch.linalg.vector_norm(v, dim=-1, keepdim=True)
    k = v / norms
    angle = norms[..., 0]
    z = torch.zeros_like(k[..., 0])
    K = torch.stack([
        torch.stack([z, -k[..., 2], k[..., 1]], dim=-1),
        torch.stack([k[..., 2], z, -k[..., 0]], dim=-1),
        torch.stack([-k[..., 1], k[..., 0], z], dim=-1),
    ], dim=-2)
    c = torch.cos(angle)[..., None, None]
    s = torch.sin(angle)[..., None, None]
    eye = torch.eye(3, dtype=v.dtype, device=v.device)
    r = c * eye + (1 - c) * k[..., :, None] * k[..., None, :] + s * K
    return r if three_by_three else pad_3x3_to_4x4(r)


def translation(x, name=None):
    """[*, 3] -> [*, 4, 4] with the displacement in the last ROW (row vectors; dirt/matrices.py:64-89)."""
    x = _t(x)
    out = torch.eye(4, dtype=x.dtype, device=x.device).expand(x.shape[:-1] + (4, 4)).clone()
    out[..., 3, :3] = x
    return out


def scale(x, name=None):
    """[*, 3] -> [*, 4, 4] diagonal scaling (dirt/matrices.py:92-108)."""
    x = _t(x)
    return torch.diag_embed(torch.cat([x, torch.ones_like(x[..., :1])], dim=-1))


def perspective_projection(near, far, right, aspect, name=None):
    """OpenGL perspective projection, camera looking down -z; parameters broadcast to [A1..An], result
    [A1..An, 4, 4] (dirt/matrices.py:111-152).  `aspect` = height / width of the viewport."""
    args = [a for a in (near, far, right, aspect) if isinstance(a, torch.Tensor)]
    like = args[0] if args else None
    near, far, right, aspect = (_t(a, like) for a in (near, far, right, aspect))
    top = right * aspect
    near, far, right, top = torch.broadcast_tensors(near, far, right, top)
    out = near.new_zeros(near.shape + (4, 4))  # indexed [*, in, out]
    out[..., 0, 0] = near / right
    out[..., 1, 1] = near / top
    out[..., 2, 2] = -(far + near) / (far - near)
    out[..., 3, 2] = -2. * far * near / (far - near)
    out[..., 2, 3] = -1.
    return out


def compose(*matrices):
    """Product of the given transforms, the first applied first (dirt/matrices.py:183-207)."""
    if len(matrices) == 0:
        return torch.eye(4)
    result = _t(matrices[0])
    for m in matrices[1:]:
        result = torch.matmul(result, _t(m, result))
    return result
