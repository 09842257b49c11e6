"""Synthetic scene generators for tests, smoke and bench (numpy only).

K1 = the quad of the reference's tests/square_test.py:20-36; K2 = the cube of samples/simple.py:15-66;
K3/K4/K5 = `rand_mesh` scenes.  Everything is deterministic in its seed.
"""
import math
import numpy as np


def square_scene():
    """tests/square_test.py:6-8,20-36 verbatim: 128x128x1, 2-triangle 16x16 square centred at (32, 64)."""
    w = h = 128
    cx, cy, size = 32, 64, 16
    v = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], np.float32) * size - size / 2.
    v = v + np.array([cx, cy], np.float32)
    v = v * 2. / np.array([w, h], np.float32) - 1.
    vertices = np.concatenate([v, np.zeros([4, 1], np.float32), np.ones([4, 1], np.float32)], axis=1).astype(np.float32)
    faces = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return dict(background=np.zeros([h, w, 1], np.float32), vertices=vertices,
                vertex_colors=np.ones([4, 1], np.float32), faces=faces, height=h, width=w, channels=1)


def square_expected():
    """tests/square_test.py:11-17: the analytic mask the reference compares against."""
    w = h = 128
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    xs = xs.astype(np.float32) + 0.5
    ys = ys.astype(np.float32) + 0.5
    return ((np.abs(xs - 32) <= 8) & (np.abs(ys - 64) <= 8)).astype(np.float32)


def rand_mesh(face_count, seed, r_lo, r_hi, shared=False):
    """SURVEY.md 8d `rand_mesh`: random perspective triangles, all w > 0.

    Returns (vertices [V,4] clip space float32, faces [F,3] int32).  `shared=True` gives the
    shared-vertex variant (a jittered grid, V ~ F/2) that stresses atomics."""
    rng = np.random.default_rng(seed)
    F = int(face_count)
    if not shared:
        c = rng.uniform(-1, 1, [F, 1, 2])
        r = rng.uniform(r_lo, r_hi, [F, 1, 1])
        th0 = rng.uniform(0, 2 * math.pi, [F, 1])
        th = th0 + 2 * math.pi * np.arange(3)[None, :] / 3 + rng.uniform(-0.5, 0.5, [F, 3])
        ndc_xy = c + r * np.stack([np.cos(th), np.sin(th)], axis=-1)
        ndc_z = rng.uniform(-0.9, 0.9, [F, 3, 1])
        d = rng.uniform(1, 4, [F, 1, 1])
        w = d * (1 + rng.uniform(-0.1, 0.1, [F, 3, 1]))
        verts = np.concatenate([ndc_xy * w, ndc_z * w, w], axis=-1).reshape(F * 3, 4).astype(np.float32)
        faces = np.arange(F * 3, dtype=np.int32).reshape(F, 3)
        return verts, faces
    n = max(2, int(math.ceil(math.sqrt(F / 2.0))) + 1)
    gx, gy = np.meshgrid(np.linspace(-1.05, 1.05, n), np.linspace(-1.05, 1.05, n))
    jit = rng.uniform(-0.3, 0.3, [n, n, 2]) * (2.1 / (n - 1))
    ndc_xy = np.stack([gx, gy], -1) + jit
    ndc_z = rng.uniform(-0.9, 0.9, [n, n, 1])
    w = rng.uniform(1, 4, [n, n, 1])
    verts = np.concatenate([ndc_xy * w, ndc_z * w, w], axis=-1).reshape(n * n, 4).astype(np.float32)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c_, d_ = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    faces = np.concatenate([np.stack([a, b, c_], 1), np.stack([a, c_, d_], 1)], 0)
    faces = faces[rng.permutation(len(faces))][:F].astype(np.int32)
    return verts, faces


def rand_scene(face_count, height, width, channels, seed, r_lo=0.005, r_hi=0.04, shared=False):
    """A full single scene: mesh + U(0,1) colours/background + N(0,1) grad_pixels."""
    verts, faces = rand_mesh(face_count, seed, r_lo, r_hi, shared)
    rng = np.random.default_rng(seed + 100003)
    V = verts.shape[0]
    return dict(
        background=rng.uniform(0, 1, [height, width, channels]).astype(np.float32),
        vertices=verts, faces=faces,
        vertex_colors=rng.uniform(0, 1, [V, channels]).astype(np.float32),
        grad_pixels=rng.standard_normal([height, width, channels]).astype(np.float32),
        height=height, width=width, channels=channels)


def batch_scene(face_count, height, width, channels, seeds, **kw):
    """Stack `rand_scene`s along a leading batch dimension (K4)."""
    ss = [rand_scene(face_count, height, width, channels, s, **kw) for s in seeds]
    out = {k: np.stack([s[k] for s in ss]) for k in ('background', 'vertices', 'faces', 'vertex_colors', 'grad_pixels')}
    out.update(height=height, width=width, channels=channels)
    return out


CONFIGS = {
    # name: (F, H, W, C, seed, r_lo, r_hi)   -- BASELINE.md section 3
    'K3': (10000, 1024, 1024, 4, 0, 0.005, 0.04),
    'K3-256': (10000, 256, 256, 4, 0, 0.005, 0.04),
    'K3-2048': (10000, 2048, 2048, 4, 0, 0.005, 0.04),
    'K3-384': (10000, 384, 384, 4, 0, 0.005, 0.04),     # in-between frame sizes: where the kernels' tile shapes switch
    'K3-512': (10000, 512, 512, 4, 0, 0.005, 0.04),
    'K3-768': (10000, 768, 768, 4, 0, 0.005, 0.04),
    'K3-3ch': (10000, 1024, 1024, 3, 0, 0.005, 0.04),    # the headline mesh with an RGB / a one-channel image (the reference op's own channel counts)
    'K3-1ch': (10000, 1024, 1024, 1, 0, 0.005, 0.04),
    'K2-cube': (12, 256, 256, 3, 0, 0.2, 0.6),           # a dozen large triangles at K2's frame (the cube itself: scenes.cube_scene)
    'K5': (50000, 2048, 2048, 16, 1, 0.002, 0.018),
    'K5-3ch': (50000, 2048, 2048, 3, 1, 0.002, 0.018),   # K5's mesh with an RGB image (what deferred shading filters)
}


def config_scene(name):
    F, H, W, C, seed, r_lo, r_hi = CONFIGS[name]
    return rand_scene(F, H, W, C, seed, r_lo, r_hi)


def _rodrigues(v):
    v = np.asarray(v, np.float64) + 1e-12
    n = np.linalg.norm(v)
    k = v / n
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = math.cos(n) * np.eye(3) + (1 - math.cos(n)) * np.outer(k, k) + math.sin(n) * K
    M = np.eye(4)
    M[:3, :3] = R
    return M


def cube_scene(height=256, width=256):
    """K2: the Gouraud cube of samples/simple.py:15-66 (split vertices, lit colours), numpy restatement."""
    verts = np.array([[x, y, z] for z in [-1, 1] for y in [-1, 1] for x in [-1, 1]], np.float64)
    quads = [[0, 1, 3, 2], [4, 5, 7, 6], [1, 5, 4, 0], [2, 6, 7, 3], [4, 6, 2, 0], [3, 7, 5, 1]]
    tris = np.array(sum([[[a, b, c], [c, d, a]] for a, b, c, d in quads], []), np.int32)
    v = verts[tris.reshape(-1)]                                   # split_vertices_by_face
    faces = np.arange(len(v), dtype=np.int32).reshape(-1, 3)
    v = np.concatenate([v, np.ones([len(v), 1])], 1)
    world = v @ _rodrigues([0., 0.5, 0.])
    tri = world[:, :3].reshape(-1, 3, 3)
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    n /= (np.linalg.norm(n, axis=-1, keepdims=True) + 1e-12)
    normals = np.repeat(n, 3, axis=0)
    T = np.eye(4)
    T[3, :3] = [0., -1.5, -3.5]
    view = T @ _rodrigues([-0.3, 0., 0.])
    cam = world @ view
    near, far, right = 0.1, 20., 0.1
    top = right * float(height) / width
    Pm = np.array([[near / right, 0, 0, 0], [0, near / top, 0, 0],
                   [0, 0, -(far + near) / (far - near), -2. * far * near / (far - near)], [0, 0, -1., 0]]).T
    clip = cam @ Pm
    light = np.array([1., 0., 0.])
    light /= np.linalg.norm(light)
    diffuse = np.abs(normals @ -light)[:, None] * np.ones([1, 3])
    colors = diffuse * 0.8 + 0.2
    return dict(background=np.zeros([height, width, 3], np.float32), vertices=clip.astype(np.float32),
                vertex_colors=colors.astype(np.float32), faces=faces, height=height, width=width, channels=3)


def hostile_scene(H, W, C, seed, n_small=200):
    """Geometry that exercises every special case of the set-up / coverage / depth rules at once (the cases the
    oracle's own tests check one by one): triangles crossing w = 0 and the near / far planes, a triangle wholly
    behind the eye, zero-area and repeated-vertex faces, an out-of-range and a negative vertex index, a NaN and
    an inf vertex, frame-filling triangles (the binning "big" list), coplanar duplicates (depth ties: the
    earlier face wins), a face with enormous coordinates, and a cluster of sub-pixel triangles (more faces in
    one tile than the gradient kernel's slot table holds)."""
    rng = np.random.default_rng(seed)
    verts, faces = [], []

    def add(tri):  # tri: 3 x (x, y, z, w)
        base = len(verts)
        verts.extend(tri)
        faces.append((base, base + 1, base + 2))

    # ordinary random triangles
    mv, mf = rand_mesh(n_small // 2, seed + 100, 0.05, 0.4)
    for f in mf:
        add([tuple(mv[i]) for i in f])
    # crossing w = 0 / near plane: one or two vertices behind the eye
    for _ in range(8):
        tri = []
        for k in range(3):
            w = rng.uniform(-1.0, 2.0) if k < 2 else rng.uniform(0.5, 2.0)
            tri.append((rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), rng.uniform(-1.2, 1.2) * abs(w), w))
        add(tri)
    add([(0.1, 0.2, 0.0, -1.0), (0.5, -0.2, 0.0, -2.0), (-0.3, 0.1, 0.0, -0.5)])      # wholly behind the eye
    add([(-0.5, -0.5, 1.5, 1.0), (0.5, -0.5, 1.5, 1.0), (0.0, 0.5, 1.5, 1.0)])          # beyond the far plane
    add([(-0.9, -0.9, -1.5, 1.0), (0.9, -0.9, 0.5, 1.0), (0.0, 0.9, 0.5, 1.0)])          # cut by the near plane
    add([(0.2, 0.2, 0.0, 1.0), (0.2, 0.2, 0.0, 1.0), (0.6, 0.1, 0.0, 1.0)])              # repeated vertex
    add([(-0.4, -0.4, 0.0, 1.0), (0.0, 0.0, 0.0, 1.0), (0.4, 0.4, 0.0, 1.0)])            # collinear
    add([(np.nan, 0.0, 0.0, 1.0), (0.3, 0.3, 0.0, 1.0), (0.1, 0.5, 0.0, 1.0)])           # NaN
    add([(np.inf, 0.0, 0.0, 1.0), (0.3, -0.3, 0.0, 1.0), (0.1, -0.5, 0.0, 1.0)])         # inf
    add([(-3.0, -3.0, 0.8, 1.0), (3.0, -3.0, 0.8, 1.0), (0.0, 3.0, 0.8, 1.0)])            # fills the frame, far
    add([(-1.0, -1.0, 0.6, 1.0), (1.0, -1.0, 0.6, 1.0), (-1.0, 1.0, 0.6, 1.0)])           # half the frame, edges on the border
    add([(1.0, -1.0, 0.6, 1.0), (1.0, 1.0, 0.6, 1.0), (-1.0, 1.0, 0.6, 1.0)])             # ... its watertight partner
    add([(-1e6, -1e6, 0.7, 1.0), (1e6, -1e6, 0.7, 1.0), (0.0, 1e6, 0.7, 1.0)])            # enormous
    t = [(-0.2, -0.6, 0.1, 1.0), (0.6, -0.5, 0.1, 1.0), (0.1, 0.1, 0.1, 1.0)]
    add(t); add(t)                                                                        # coplanar duplicates: exact depth tie
    # a cluster of tiny triangles inside one tile
    cx, cy = rng.uniform(-0.5, 0.5, 2)
    for _ in range(n_small // 2):
        c = np.array([cx, cy]) + rng.uniform(-24.0 / W, 24.0 / W, 2)
        r = rng.uniform(1.5, 4.0) / W
        a0 = rng.uniform(0, 2 * np.pi)
        z, w = rng.uniform(-0.5, 0.5), rng.uniform(1.0, 2.0)
        add([((c[0] + r * np.cos(a0 + 2.1 * k)) * w, (c[1] + r * np.sin(a0 + 2.1 * k)) * w, z * w, w) for k in range(3)])
    faces.append((0, 1, len(verts) + 5))   # index out of range
    faces.append((2, -1, 3))               # negative index
    order = rng.permutation(len(faces))    # draw order matters for ties only; shuffle everything else
    vertices = np.asarray(verts, np.float32)
    faces = np.asarray(faces, np.int32)[order]
    V = vertices.shape[0]
    return {'background': rng.uniform(0, 1, (H, W, C)).astype(np.float32), 'vertices': vertices,
            'vertex_colors': rng.uniform(0, 1, (V, C)).astype(np.float32), 'faces': faces,
            'grad_pixels': rng.standard_normal((H, W, C)).astype(np.float32),
            'height': H, 'width': W, 'channels': C}


def _perspective(near, far, right, aspect):
    """dirt/matrices.py:110-153 `perspective_projection` (OpenGL convention, row vectors), numpy restatement."""
    top = right * aspect
    return np.array([[near / right, 0, 0, 0], [0, near / top, 0, 0],
                     [0, 0, -(far + near) / (far - near), -2. * far * near / (far - near)], [0, 0, -1., 0]]).T


def make_cylinder(radius, height, end_offset, bevel, segments):
    """The bevelled cylinder of the reference's tests/rasterise_tests.py:11-47 (centred on the origin, axis along y):
    returns (vertices [4 * segments + 2, 3], faces [8 * segments, 3])."""
    angles = np.linspace(0., 2 * math.pi, segments, endpoint=False, dtype=np.float32)
    xz = np.stack([np.cos(angles), np.sin(angles)], axis=1) * radius
    ones = np.ones(segments)
    top_bevel = np.stack([xz[:, 0] * (1. - bevel), ones * -height / 2. - radius * bevel, xz[:, 1] * (1. - bevel)], axis=1)
    top = np.stack([xz[:, 0], ones * -height / 2., xz[:, 1]], axis=1)
    bottom = np.stack([xz[:, 0], ones * height / 2., xz[:, 1]], axis=1)
    bottom_bevel = np.stack([xz[:, 0] * (1. - bevel), ones * height / 2. + radius * bevel, xz[:, 1] * (1. - bevel)], axis=1)
    ends = [[0., -height / 2. - end_offset, 0.], [0., height / 2. + end_offset, 0.]]
    vertices = np.concatenate([top_bevel, top, bottom, bottom_bevel, ends], axis=0)
    faces = []
    for start in (0, segments, 2 * segments):
        for q in range(segments):
            u1, u2 = start + q, start + (q + 1) % segments
            l1, l2 = u1 + segments, u2 + segments
            faces.extend([[u1, u2, l1], [l1, u2, l2]])
    for t1 in range(segments):
        t2 = (t1 + 1) % segments
        b1 = t1 + segments * 3
        b2 = (b1 + 1) % segments      # (sic: tests/rasterise_tests.py:41)
        faces.extend([[segments * 4, t1, t2], [segments * 4 + 1, b1, b2]])
    return vertices, np.array(faces, dtype=np.int32)


def cylinder_scene(translation=(0., 0., -0.25), rotation_xy=0., bgcolor=(0.4, 0.2, 0.2), vertex_color=(0.7, 0.3, 0.6), seed=0):
    """The scene of the reference's tests/rasterise_tests.py:50-99: the cylinder `make_cylinder(0.2, 0.75, 0.1, 0., 10)`
    split to 240 vertices, rotated about z and scaled by 0.5, translated, under `perspective_projection(0.1, 20., 0.2,
    h / w)`, at 48 x 36 x 3; the first 75 vertices `vertex_color`, the rest random; the top half of the background
    `bgcolor`, the bottom half white.  The arguments are the values the reference feeds (:115)."""
    w, h = 48, 36
    verts, faces = make_cylinder(0.2, 0.75, 0.1, 0., 10)
    verts = np.concatenate([verts, np.ones([len(verts), 1])], axis=1)
    verts = verts[faces.reshape(-1)]                                    # split_vertices_by_face (dirt/lighting.py:136-179)
    faces = np.arange(len(verts), dtype=np.int32).reshape(-1, 3)
    c, s = math.cos(rotation_xy), math.sin(rotation_xy)
    view1 = np.array([[0.5 * c, -0.5 * s, 0, 0], [0.5 * s, 0.5 * c, 0, 0], [0, 0, 0.5, 0], [0, 0, 0, 1.]])
    view2 = np.eye(4)
    view2[3, :3] = translation
    clip = verts @ view1 @ view2 @ _perspective(0.1, 20., 0.2, float(h) / w)
    rng = np.random.default_rng(seed)
    V = len(verts)
    colors = np.concatenate([np.tile(np.asarray(vertex_color)[None], [75, 1]), rng.uniform(size=[V - 75, 3])], axis=0)
    bg = np.concatenate([np.tile(np.asarray(bgcolor)[None, None], [h // 2, w, 1]), np.ones([h // 2, w, 3])], axis=0)
    return dict(background=bg.astype(np.float32), vertices=clip.astype(np.float32), vertex_colors=colors.astype(np.float32),
                faces=faces, grad_pixels=rng.standard_normal([h, w, 3]).astype(np.float32), height=h, width=w, channels=3)


def cylinder_batch_scene(seed=0):
    """The batch of two of tests/rasterise_tests.py:89,123-132: the same geometry (translation (0, 0, -1), rotation 0.5)
    over a black and a blue background, random vertex colours per scene."""
    a = cylinder_scene((0., 0., -1.), 0.5, seed=seed)
    rng = np.random.default_rng(seed + 7)
    V = a['vertices'].shape[0]
    h, w = a['height'], a['width']
    bg = np.tile(np.array([[0., 0., 0.], [0., 0., 1.]], np.float32)[:, None, None, :], [1, h, w, 1])
    return dict(background=bg, vertices=np.tile(a['vertices'][None], [2, 1, 1]), faces=np.tile(a['faces'][None], [2, 1, 1]),
                vertex_colors=rng.uniform(size=[2, V, 3]).astype(np.float32),
                grad_pixels=rng.standard_normal([2, h, w, 3]).astype(np.float32), height=h, width=w, channels=3)


def bent_square_geometry(translation=(0., 0., 0.), rotation=0.5, scale=(1., 1., 1.)):
    """The bent square of the reference's tests/deferred_grad_test.py:18-55 (32 x 32 canvas): two faces split to six
    vertices, rotated about z (rodrigues), scaled, translated, moved away from the camera and projected.  Returns
    (vertices_clip [6,4], faces [2,3], vertices_world [6,4], vertex_colours [6,3]); the values are the reference's
    initial variables (rotation 0.5 as in :176-180)."""
    square_size = 4.
    v = np.array([[-1, -1, 0.], [-1, 1, 0], [1, 1, 0], [1, -1, -1.3]], np.float64) * square_size / 2
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    v = v[f.reshape(-1)]
    f = np.arange(6, dtype=np.int32).reshape(2, 3)
    v = np.concatenate([v, np.ones([6, 1])], axis=1)
    world = v @ _rodrigues([0., 0., rotation]) * np.concatenate([scale, [1.]]) + np.concatenate([translation, [0.]])
    view = np.eye(4)
    view[3, :3] = [-0.5, 0., -3.5]
    clip = world @ view @ _perspective(0.1, 20., 0.1, 1.0)
    colours = np.concatenate([np.ones([3, 3]) * [0.8, 0.5, 0.], np.ones([3, 3]) * [0.5, 0.8, 0.]], axis=0)
    return clip.astype(np.float32), f, world.astype(np.float32), colours.astype(np.float32)
