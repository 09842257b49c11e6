#!/usr/bin/env python3
"""The textured, deferred-shaded cube of the reference's samples/textured.py on dirt_amd (MI355X), with a
procedural texture instead of cat.jpg.  Renders a G-buffer of (mask, uv, normal), shades it with a texture
look-up + diffuse lighting, and back-propagates an image loss to the texture, the light direction and the
vertices.  Writes textured.png next to this file when Pillow is available.

    python examples/textured.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dirt_amd as dirt  # noqa: E402
from dirt_amd import lighting, matrices, texture as tex  # noqa: E402

frame_width, frame_height = 640, 480


def build_cube():
    vertices, uvs, faces = [], [], []

    def add_quad(v, uv):
        index = len(vertices)
        faces.extend([[index + 2, index + 1, index], [index, index + 3, index + 2]])
        vertices.extend(v)
        uvs.extend(uv)

    add_quad([[-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], [[0.1, 0.9], [0.9, 0.9], [0.9, 0.1], [0.1, 0.1]])
    add_quad([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1]], [[1, 1], [0, 1], [0, 0], [1, 0]])
    add_quad([[1, 1, 1], [1, 1, -1], [1, -1, -1], [1, -1, 1]], [[0.3, 0.25], [0.6, 0.25], [0.6, 0.55], [0.3, 0.55]])
    add_quad([[-1, 1, 1], [-1, 1, -1], [-1, -1, -1], [-1, -1, 1]], [[0.4, 0.4], [0.5, 0.4], [0.5, 0.5], [0.4, 0.5]])
    add_quad([[-1, 1, -1], [1, 1, -1], [1, 1, 1], [-1, 1, 1]], [[0, 0], [2, 0], [2, 2], [0, 2]])
    add_quad([[-1, -1, -1], [1, -1, -1], [1, -1, 1], [-1, -1, 1]], [[0, 0], [2, 0], [2, 2], [0, 2]])
    return np.asarray(vertices, np.float32), np.asarray(uvs, np.float32), np.asarray(faces, np.int32)


def checker_texture(size=128):
    y, x = np.mgrid[0:size, 0:size]
    c = ((x // 16 + y // 16) % 2).astype(np.float32)
    return np.stack([0.2 + 0.8 * c, 0.3 + 0.5 * (x / size), 0.9 - 0.6 * c], -1).astype(np.float32)


def shader_fn(gbuffer, texture, light_direction):
    mask, uvs, normals = gbuffer[..., :1], gbuffer[..., 1:3], gbuffer[..., 3:]
    unlit = tex.sample_texture_uv(texture, uvs)  # one kernel: uvs_to_pixel_indices + bilinear sample_texture (samples/textured.py:16-61)
    ambient = unlit * 0.4
    diffuse = lighting.diffuse_directional(normals.reshape(-1, 3), unlit.reshape(-1, 3), light_direction,
                                           light_color=torch.full((3,), 0.6, device=gbuffer.device), double_sided=True)
    background = torch.tensor([0., 0., 0.3], device=gbuffer.device)
    return (diffuse.reshape(unlit.shape) + ambient) * mask + background * (1. - mask)


def geometry(vertices_object, uvs, faces):
    """-> (clip-space vertices [V,4], vertex attributes [V,6] = mask, texture coordinates, normals)."""
    device = vertices_object.device
    v = torch.cat([vertices_object, torch.ones_like(vertices_object[:, -1:])], dim=1)
    world = v @ matrices.rodrigues(torch.tensor([0., 0.6, 0.], device=device))
    normals = lighting.vertex_normals(world, faces)
    view = matrices.compose(matrices.translation(torch.tensor([0., -2., -3.2], device=device)),
                            matrices.rodrigues(torch.tensor([-0.5, 0., 0.], device=device)))
    clip = (world @ view) @ matrices.perspective_projection(near=0.1, far=20., right=0.1, aspect=float(frame_height) / frame_width).to(device)
    return clip, torch.cat([torch.ones_like(v[:, :1]), uvs, normals], dim=1)


def render(vertices_object, uvs, faces, texture, light_direction):
    clip, attributes = geometry(vertices_object, uvs, faces)
    return dirt.rasterise_deferred(
        vertices=clip, vertex_attributes=attributes, faces=faces,
        background_attributes=torch.zeros([frame_height, frame_width, 6], device=clip.device),
        shader_fn=shader_fn, shader_additional_inputs=[texture, light_direction])


def main():
    device = torch.device('cuda', 0)
    vertices, uvs, faces = (torch.from_numpy(a).to(device) for a in build_cube())
    texture = torch.from_numpy(checker_texture()).to(device).requires_grad_(True)
    light = torch.nn.functional.normalize(torch.tensor([1., -0.3, -0.5], device=device), dim=0).requires_grad_(True)
    vertices.requires_grad_(True)
    pixels = render(vertices, uvs, faces, texture, light)
    (pixels ** 2).mean().backward()
    print('pixels', tuple(pixels.shape), 'mean %.4f' % pixels.mean().item())
    print('|d loss / d texture| max %.3e, |d loss / d light| %s, |d loss / d vertices| max %.3e'
          % (texture.grad.abs().max().item(), light.grad.abs().cpu().numpy().round(5), vertices.grad.abs().max().item()))
    try:
        from PIL import Image
        Image.fromarray((pixels.detach().clamp(0, 1) * 255).byte().cpu().numpy()).save(
            os.path.join(os.path.dirname(os.path.abspath(__file__)), 'textured.png'))
    except ImportError:
        pass


if __name__ == '__main__':
    main()
