#!/usr/bin/env python3
"""Direct against deferred shading, pixel by pixel: the experiment of the reference's tests/deferred_grad_test.py on
`import dirt` (torch tensors in place of TensorFlow ones).

A bent square (two faces, six split vertices) is placed by five variables -- translation [3], rotation about z, scale, the
light's intensity, the background colour [3] -- and rendered twice at 32 x 32: lit per VERTEX and rasterised
(`dirt.rasterise`), and as a 7-channel G-buffer (mask, colour, normal) lit per PIXEL (`dirt.rasterise_deferred`).  For both
routes the full Jacobian d pixel / d variable is formed, one backward pass per pixel and channel (the reference's
`get_pixel_gradients`, tests/deferred_grad_test.py:198-216), and the two are compared and written side by side as images
(Pillow permitting).  The reference only looks at the images; `main()` also returns the numbers a test can assert.

    python examples/deferred_jacobians.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dirt  # noqa: E402
from dirt import lighting, matrices  # noqa: E402

SIDE = 32            # canvas (tests/deferred_grad_test.py:8)
SQUARE = 4.          # edge of the square in object units (:9)
NAMES = ['translation', 'rotation', 'scale', 'light_intensity', 'background']


def place_geometry(translation, rotation, scale, device):
    """Object -> world -> camera -> clip, as tests/deferred_grad_test.py:18-55."""
    corners = torch.tensor([[-1, -1, 0.], [-1, 1, 0], [1, 1, 0], [1, -1, -1.3]], device=device) * (SQUARE / 2)
    corners, faces = lighting.split_vertices_by_face(corners, torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32, device=device))
    homogeneous = torch.cat([corners, torch.ones_like(corners[:, :1])], dim=1)
    axis = torch.stack([rotation * 0, rotation * 0, rotation])
    world = homogeneous @ matrices.rodrigues(axis) * scale + torch.cat([translation, translation.new_zeros(1)])
    normals = lighting.vertex_normals(world, faces)
    camera = world @ matrices.translation(torch.tensor([-0.5, 0., -3.5], device=device))
    clip = camera @ matrices.perspective_projection(near=0.1, far=20., right=0.1, aspect=1.).to(device)
    tint = torch.tensor([[0.8, 0.5, 0.]] * 3 + [[0.5, 0.8, 0.]] * 3, device=device)
    return clip, faces, normals, tint


def lit(colours, normals, intensity):
    """Ambient + one green directional light (tests/deferred_grad_test.py:58-70), for vertices or for G-buffer pixels."""
    toward = torch.nn.functional.normalize(torch.tensor([1., -0.3, -0.5], device=colours.device), dim=0)
    green = torch.tensor([0., 1., 0.], device=colours.device) * intensity
    diffuse = lighting.diffuse_directional(normals.reshape(-1, 3), colours.reshape(-1, 3), toward, light_color=green, double_sided=True)
    return colours * 0.4 + diffuse.reshape(colours.shape)


def render_direct(clip, faces, normals, tint, intensity, background):
    return dirt.rasterise(vertices=clip, faces=faces, vertex_colors=lit(tint, normals, intensity),
                          background=torch.ones(SIDE, SIDE, 3, device=clip.device) * background)


def render_deferred(clip, faces, normals, tint, intensity, background):
    attributes = torch.cat([torch.ones_like(clip[:, :1]), tint, normals], dim=1)

    def shade(gbuffer, intensity_, background_):
        mask, colours, nrm = gbuffer.split([1, 3, 3], dim=-1)
        return mask * lit(colours, nrm, intensity_) + (1. - mask) * background_

    return dirt.rasterise_deferred(torch.zeros(SIDE, SIDE, 7, device=clip.device), clip, attributes, faces, shade, [intensity, background])


def jacobian(pixels, variables):
    """[SIDE, SIDE, 3, 9]: one backward pass per pixel and channel, the variables' gradients side by side."""
    rows = []
    flat = pixels.reshape(-1)
    for i in range(flat.numel()):
        grads = torch.autograd.grad(flat[i], variables, retain_graph=True, allow_unused=True)
        rows.append(torch.cat([(g if g is not None else torch.zeros_like(v)).reshape(-1) for g, v in zip(grads, variables)]))
    return torch.stack(rows).reshape(SIDE, SIDE, 3, -1)


def as_images(j_direct, j_deferred):
    """Both Jacobians under ONE normalisation (tests/deferred_grad_test.py:145-166): per variable a SIDE x SIDE RGB panel."""
    both = torch.stack([j_direct, j_deferred])
    lo = both.amin(dim=(0, 1, 2), keepdim=True)
    norm = (both - lo) / (both - lo).amax(dim=(0, 1, 2), keepdim=True).clamp_min(1e-30)
    return [n.permute(0, 3, 1, 2).reshape(SIDE, -1, 3) for n in norm.permute(0, 1, 2, 4, 3)]   # [SIDE, 9 * SIDE, 3] each


def main(write_images=True, device=None):
    device = device or torch.device('cuda', 0)
    variables = [torch.tensor(v, device=device, requires_grad=True) for v in ([0., 0., 0.], 0.5, 1., 0.6, [0., 0., 0.2])]
    translation, rotation, scale, intensity, background = variables
    clip, faces, normals, tint = place_geometry(translation, rotation, scale, device)
    direct = render_direct(clip, faces, normals, tint, intensity, background)
    deferred = render_deferred(clip, faces, normals, tint, intensity, background)
    j_direct, j_deferred = jacobian(direct, variables), jacobian(deferred, variables)

    covered = (direct.detach() - background.detach()).abs().amax(-1) > 1e-6
    report = {'pixels_max_abs_difference': float((direct - deferred).detach().abs().max()), 'covered_pixels': int(covered.sum())}
    # per variable: the two routes' total sensitivity (sum over pixels of |d pixel / d variable|) and how far apart they are
    widths = [v.numel() for v in variables]
    at = 0
    for name, wd in zip(NAMES, widths):
        a, b = j_direct[..., at:at + wd].detach(), j_deferred[..., at:at + wd].detach()
        report[name] = {'direct_l1': float(a.abs().sum()), 'deferred_l1': float(b.abs().sum()),
                        'max_abs_difference': float((a - b).abs().max())}
        at += wd
    if write_images:
        try:
            from PIL import Image
            here = os.path.dirname(os.path.abspath(__file__))
            def save(name, t):
                Image.fromarray((t.detach().clamp(0, 1) * 255).byte().cpu().numpy()).save(os.path.join(here, name))
            save('deferred_jacobians_pixels.png', torch.cat([direct, deferred], dim=1))
            panels = as_images(j_direct.detach(), j_deferred.detach())
            save('deferred_jacobians_grads.png', torch.cat(panels, dim=0))
        except ImportError:
            pass
    return report


if __name__ == '__main__':
    out = main()
    print('pixels: direct vs deferred max |difference| %.3g over %d covered pixels' % (out['pixels_max_abs_difference'], out['covered_pixels']))
    for name in NAMES:
        print('%-16s sum |d pixel / d variable|: direct %10.4f  deferred %10.4f   max |difference| %.4f'
              % (name, out[name]['direct_l1'], out[name]['deferred_l1'], out[name]['max_abs_difference']))
