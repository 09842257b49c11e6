"""The reference's tests/square_test.py (its only known-answer test) as a runnable script on this implementation:

    python tests/square_test.py        ->  successful: all pixels agree

Same construction and the same final comparison as the reference (:11-17 analytic mask, :20-36 the square through
`dirt.rasterise`, :54-57 exact equality), with torch tensors on the GPU in place of TensorFlow ones.  Needs an MI355X
(the reference needs an NVIDIA GPU); under pytest it is a `gpu` test."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dirt  # noqa: E402

canvas_width, canvas_height = 128, 128
centre_x, centre_y = 32, 64
square_size = 16


def get_non_dirt_pixels():
    ys, xs = torch.meshgrid(torch.arange(canvas_height), torch.arange(canvas_width), indexing='ij')
    xs = xs.to(torch.float32) + 0.5
    ys = ys.to(torch.float32) + 0.5
    x_in_range = (xs - centre_x).abs() <= square_size / 2
    y_in_range = (ys - centre_y).abs() <= square_size / 2
    return (x_in_range & y_in_range).to(torch.float32)


def get_dirt_pixels(device='cuda:0'):
    # the square in screen space, then homogeneous clip-space coordinates
    corners = torch.tensor([[0, 0], [0, 1], [1, 1], [1, 0]], dtype=torch.float32, device=device)
    square_vertices = corners * square_size - square_size / 2. + torch.tensor([centre_x, centre_y], dtype=torch.float32, device=device)
    square_vertices = square_vertices * 2. / torch.tensor([canvas_width, canvas_height], dtype=torch.float32, device=device) - 1.
    square_vertices = torch.cat([square_vertices, torch.zeros([4, 1], device=device), torch.ones([4, 1], device=device)], dim=1)
    return dirt.rasterise(
        vertices=square_vertices,
        faces=[[0, 1, 2], [0, 2, 3]],
        vertex_colors=torch.ones([4, 1], device=device),
        background=torch.zeros([canvas_height, canvas_width, 1], device=device),
        height=canvas_height, width=canvas_width, channels=1
    )[:, :, 0]


def main():
    non_dirt_pixels = get_non_dirt_pixels().numpy()
    dirt_pixels = get_dirt_pixels().cpu().numpy()
    if np.all(non_dirt_pixels == dirt_pixels):
        print('successful: all pixels agree')
        return 0
    print('failed: {} pixels disagree'.format(np.sum(non_dirt_pixels != dirt_pixels)))
    return 1


@pytest.mark.gpu
def test_square_script_reports_success(gpu, capsys):
    assert main() == 0
    assert 'successful: all pixels agree' in capsys.readouterr().out


if __name__ == '__main__':
    sys.exit(main())
