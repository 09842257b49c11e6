#!/usr/bin/env python3
"""The deferred-shaded cube of the reference's samples/deferred.py on dirt_amd (MI355X): a 10-channel G-buffer (mask,
world position, colour, normal) rendered once and shaded per pixel -- ambient + red diffuse + white Phong specular, with
the view matrix and the light direction as `shader_additional_inputs` (samples/deferred.py:58-117) -- then an image loss
back-propagated to the vertices, the view matrix and the light direction.  Writes deferred.png next to this file when
Pillow is available.

    python examples/deferred.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dirt_amd as dirt  # noqa: E402
from dirt_amd import lighting, matrices  # noqa: E402

frame_width, frame_height = 640, 480


def build_cube():   # samples/deferred.py:13-21
    vertices = [[x, y, z] for z in [-1, 1] for y in [-1, 1] for x in [-1, 1]]
    quads = [[0, 1, 3, 2], [4, 5, 7, 6], [1, 5, 4, 0], [2, 6, 7, 3], [4, 6, 2, 0], [3, 7, 5, 1]]
    triangles = sum([[[a, b, c], [c, d, a]] for [a, b, c, d] in quads], [])
    return np.asarray(vertices, np.float32), np.asarray(triangles, np.int32)


def shader_fn(gbuffer, view_matrix, light_direction):
    """samples/deferred.py:58-96: per-pixel lighting of the G-buffer."""
    mask, positions, unlit_colors, normals = gbuffer[..., :1], gbuffer[..., 1:4], gbuffer[..., 4:7], gbuffer[..., 7:]
    dev = gbuffer.device
    ambient = unlit_colors * 0.2
    diffuse = lighting.diffuse_directional(normals.reshape(-1, 3), unlit_colors.reshape(-1, 3), light_direction,
                                           light_color=torch.tensor([1., 0., 0.], device=dev), double_sided=False)
    camera_position_world = torch.linalg.inv(view_matrix)[3, :3]
    specular = lighting.specular_directional(positions.reshape(-1, 3), normals.reshape(-1, 3), unlit_colors.reshape(-1, 3),
                                             light_direction, light_color=torch.tensor([1., 1., 1.], device=dev),
                                             camera_position=camera_position_world, shininess=6., double_sided=False)
    lit = diffuse.reshape(unlit_colors.shape) + specular.reshape(unlit_colors.shape) + ambient
    return torch.clamp(lit * mask + torch.tensor([0., 0., 0.3], device=dev) * (1. - mask), 0., 1.)


def geometry(vertices_object, faces, view_matrix):
    """-> (clip-space vertices [36,4], faces [12,3], vertex attributes [36,10] = mask, world positions, colours, normals)."""
    dev = vertices_object.device
    vertices_object, faces = lighting.split_vertices_by_face(vertices_object, faces)
    colors = torch.ones_like(vertices_object)
    v = torch.cat([vertices_object, torch.ones_like(vertices_object[:, -1:])], dim=1)
    world = v @ matrices.rodrigues(torch.tensor([0., 0.5, 0.], device=dev))
    normals = lighting.vertex_normals_pre_split(world, faces)
    clip = (world @ view_matrix) @ matrices.perspective_projection(near=0.1, far=20., right=0.1, aspect=float(frame_height) / frame_width).to(dev)
    return clip, faces, torch.cat([torch.ones_like(v[:, :1]), world[:, :3], colors, normals], dim=1)


def render(vertices_object, faces, view_matrix, light_direction):
    clip, faces, attributes = geometry(vertices_object, faces, view_matrix)
    return dirt.rasterise_deferred(
        vertices=clip, vertex_attributes=attributes, faces=faces,
        background_attributes=torch.zeros([frame_height, frame_width, 10], device=clip.device),
        shader_fn=shader_fn, shader_additional_inputs=[view_matrix, light_direction])


def main():
    dev = torch.device('cuda', 0)
    vertices, faces = (torch.from_numpy(a).to(dev) for a in build_cube())
    vertices.requires_grad_(True)
    view_matrix = matrices.compose(matrices.translation(torch.tensor([0., -1.5, -3.5], device=dev)),
                                   matrices.rodrigues(torch.tensor([-0.3, 0., 0.], device=dev))).requires_grad_(True)
    light = torch.nn.functional.normalize(torch.tensor([1., -0.3, -0.5], device=dev), dim=0).requires_grad_(True)
    pixels = render(vertices, faces, view_matrix, light)
    (pixels ** 2).mean().backward()
    print('pixels', tuple(pixels.shape), 'mean %.4f' % pixels.mean().item())
    print('|d loss / d vertices| max %.3e, |d loss / d view| max %.3e, d loss / d light %s'
          % (vertices.grad.abs().max().item(), view_matrix.grad.abs().max().item(), light.grad.cpu().numpy().round(5)))
    try:
        from PIL import Image
        Image.fromarray((pixels.detach() * 255).byte().cpu().numpy()).save(
            os.path.join(os.path.dirname(os.path.abspath(__file__)), 'deferred.png'))
    except ImportError:
        pass
    return pixels


if __name__ == '__main__':
    main()
