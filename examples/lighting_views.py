#!/usr/bin/env python3
"""The four views of the reference's tests/lighting_tests.py on `import dirt`: a bevelled cylinder at 256 x 192 coloured by
its normals, by a directional light, by a point light, and by the same point light on the mesh split into per-face
vertices (`split_vertices_by_face` + `vertex_normals_pre_split`).  The reference shows them with cv2.imshow; here they are
written as PNGs next to this file (Pillow permitting) and `main()` returns them for a test to compare.

    python examples/lighting_views.py
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dirt  # noqa: E402
from dirt import lighting, matrices  # noqa: E402

WIDTH, HEIGHT = 256, 192


def cylinder(radius, height, end_offset, bevel, segments):
    """Rings (bevelled top, top, bottom, bevelled bottom) + two end points: the mesh of tests/rasterise_tests.py:11-47,
    including its wrap-around of the bottom fan's second index."""
    ang = np.linspace(0., 2 * math.pi, segments, endpoint=False)
    ring = lambda scale, y: np.stack([np.cos(ang) * radius * scale, np.full(segments, y), np.sin(ang) * radius * scale], axis=1)
    half = height / 2.
    rings = [ring(1. - bevel, -half - radius * bevel), ring(1., -half), ring(1., half), ring(1. - bevel, half + radius * bevel)]
    points = np.concatenate(rings + [np.array([[0., -half - end_offset, 0.], [0., half + end_offset, 0.]])], axis=0)
    tris = []
    for band in range(3):
        for q in range(segments):
            a, b = band * segments + q, band * segments + (q + 1) % segments
            tris += [[a, b, a + segments], [a + segments, b, b + segments]]
    for q in range(segments):
        tris += [[4 * segments, q, (q + 1) % segments], [4 * segments + 1, 3 * segments + q, (3 * segments + q + 1) % segments]]
    return points.astype(np.float32), np.array(tris, np.int32)


def main(write_images=True, device=None, rotation_xy=0., translation=(0., 0., -0.25)):
    device = device or torch.device('cuda', 0)
    points, tris = cylinder(0.2, 0.75, 0.1, 0.2, 32)
    faces = torch.from_numpy(tris).to(device)
    vertices = torch.cat([torch.from_numpy(points), torch.ones(len(points), 1)], dim=1).to(device)
    c, s = math.cos(rotation_xy), math.sin(rotation_xy)
    spin = torch.tensor([[0.5 * c, -0.5 * s, 0., 0.], [0.5 * s, 0.5 * c, 0., 0.], [0., 0., 0.5, 0.], [0., 0., 0., 1.]], device=device)
    placed = vertices @ spin @ matrices.translation(torch.tensor(translation, device=device))
    normals = lighting.vertex_normals(placed[:, :3], faces)
    placed_split, faces_split = lighting.split_vertices_by_face(placed, faces)
    normals_split = lighting.vertex_normals_pre_split(placed_split[:, :3], faces_split)
    projection = matrices.perspective_projection(0.1, 20., 0.2, float(HEIGHT) / WIDTH).to(device)

    def view(verts, fcs, colours):
        return dirt.rasterise(background=torch.zeros(HEIGHT, WIDTH, 3, device=device), vertices=verts @ projection,
                              vertex_colors=colours.float(), faces=fcs, height=HEIGHT, width=WIDTH, channels=3)

    white = lambda n: torch.ones(n, 3, device=device)
    blue = torch.tensor([0., 0., 0.4], device=device)
    t = lambda *v: torch.tensor(v, device=device)
    views = {
        'normals': view(placed, faces, normals.abs()),
        # (lights in object space: they use the placed normals and positions; tests/lighting_tests.py:47-48)
        'directional': view(placed, faces, lighting.diffuse_directional(normals, white(len(points)), t(1., 0., 0.), t(1., 1., 0.), False) + blue),
        'point': view(placed, faces, lighting.diffuse_point(placed[:, :3], normals, white(len(points)), t(0.5, -1., 0.5), t(1., 0.5, 0.9), False) + blue),
        'point_split': view(placed_split, faces_split, lighting.diffuse_point(placed_split[:, :3], normals_split, white(len(placed_split)),
                                                                             t(0.5, -1., 0.5), t(1., 0.5, 0.9), False) + blue),
    }
    if write_images:
        try:
            from PIL import Image
            here = os.path.dirname(os.path.abspath(__file__))
            for name, im in views.items():
                Image.fromarray((im.clamp(0, 1) * 255).byte().cpu().numpy()).save(os.path.join(here, 'lighting_%s.png' % name))
        except ImportError:
            pass
    return views


if __name__ == '__main__':
    out = main()
    for name, im in out.items():
        print('%-12s covered %5d pixels, mean colour %s' % (name, int((im.amax(-1) > 0).sum()), [round(float(x), 4) for x in im.mean((0, 1))]))
