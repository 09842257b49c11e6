#!/usr/bin/env python3
"""The Gouraud-shaded cube of the reference's samples/simple.py, rendered with dirt_amd on an MI355X, followed by
a few steps of gradient descent on the cube's rotation to show the gradients flowing back through the rasteriser
and the matrix helpers.  Writes simple.png next to this file when Pillow is available.

    python examples/simple.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dirt_amd as dirt  # noqa: E402
from dirt_amd import lighting, matrices  # noqa: E402

frame_width, frame_height = 640, 480


def build_cube():
    vertices = [[x, y, z] for z in [-1, 1] for y in [-1, 1] for x in [-1, 1]]
    quads = [[0, 1, 3, 2], [4, 5, 7, 6], [1, 5, 4, 0], [2, 6, 7, 3], [4, 6, 2, 0], [3, 7, 5, 1]]
    return vertices, sum([[[a, b, c], [c, d, a]] for a, b, c, d in quads], [])


def render(rotation, device):
    vertices, faces = build_cube()
    vertices = torch.tensor(vertices, dtype=torch.float32, device=device)
    vertices, faces = lighting.split_vertices_by_face(vertices, torch.tensor(faces, dtype=torch.int32, device=device))
    colors = torch.ones_like(vertices)
    vertices = torch.cat([vertices, torch.ones_like(vertices[:, -1:])], dim=1)
    world = vertices @ matrices.rodrigues(rotation)
    normals = lighting.vertex_normals_pre_split(world, faces)
    view = matrices.compose(matrices.translation(torch.tensor([0., -1.5, -3.5], device=device)),
                            matrices.rodrigues(torch.tensor([-0.3, 0., 0.], device=device)))
    projection = matrices.perspective_projection(near=0.1, far=20., right=0.1, aspect=float(frame_height) / frame_width).to(device)
    clip = (world @ view) @ projection
    lit = lighting.diffuse_directional(normals, colors, light_direction=torch.tensor([1., 0., 0.], device=device),
                                       light_color=torch.tensor([1., 1., 1.], device=device)) * 0.8 + colors * 0.2
    return dirt.rasterise(vertices=clip, faces=faces, vertex_colors=lit,
                          background=torch.zeros([frame_height, frame_width, 3], device=device),
                          width=frame_width, height=frame_height, channels=3)


def main():
    device = torch.device('cuda', 0)
    target = render(torch.tensor([0., 0.5, 0.], device=device), device).detach()
    try:
        from PIL import Image
        Image.fromarray((target.clamp(0, 1) * 255).byte().cpu().numpy()).save(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'simple.png'))
    except ImportError:
        pass
    rotation = torch.tensor([0., 0.35, 0.], device=device, requires_grad=True)
    opt = torch.optim.SGD([rotation], lr=5e-7)
    for it in range(40):
        opt.zero_grad()
        loss = ((render(rotation, device) - target) ** 2).sum()
        loss.backward()
        opt.step()
        if it % 5 == 0:
            print('step %2d  loss %9.2f  rotation.y %.4f' % (it, loss.item(), rotation[1].item()))
    print('final rotation.y %.4f (target 0.5)' % rotation[1].item())


if __name__ == '__main__':
    main()
