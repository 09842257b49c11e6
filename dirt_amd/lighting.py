"""`dirt.lighting` over torch tensors (dirt/lighting.py:1-344): mesh normals and simple reflectance models.

Same signatures and conventions as the reference: `vertices` [*, V, 3|4] with any leading batch dimensions and a
topology `faces` [F, 3] shared by the batch.  Where the reference builds a sparse [F, V, ...] tensor and reduces
it (dirt/lighting.py:64-79) this uses one `index_add_` over the vertex axis -- same sums, linear memory."""
import torch


def _prepare_vertices_and_faces(vertices, faces):
    if not isinstance(vertices, torch.Tensor):
        vertices = torch.as_tensor(vertices, dtype=torch.float32)
    faces = torch.as_tensor(faces, device=vertices.device)
    assert faces.dtype in (torch.int32, torch.int64)  # dirt/lighting.py:15-17
    return vertices, faces.long()


def _get_face_normals(vertices, faces):
    # vertices [*, V, 3], faces [F, 3] -> unit normals [*, F, 3] (dirt/lighting.py:21-28)
    v0, v1, v2 = (vertices[..., faces[:, k], :] for k in range(3))
    n = torch.linalg.cross(v1 - v0, v2 - v0, dim=-1)
    return n / (torch.linalg.vector_norm(n, dim=-1, keepdim=True) + 1.e-12)


def vertex_normals(vertices, faces, name=None):
    """Per-vertex normals: the normalised sum of the unit normals of the faces using the vertex
    (dirt/lighting.py:31-89).  [*, V, 3|4] -> [*, V, 3]."""
    vertices, faces = _prepare_vertices_and_faces(vertices, faces)
    vertices = vertices[..., :3]
    assert vertices.dim() in (2, 3)  # as the reference (dirt/lighting.py:59)
    normals_by_face = _get_face_normals(vertices, faces)  # [*, F, 3]
    summed = torch.zeros_like(vertices)
    for k in range(3):
        summed = summed.index_add(-2, faces[:, k], normals_by_face)
    return summed / (torch.linalg.vector_norm(summed, dim=-1, keepdim=True) + 1.e-12)


def vertex_normals_pre_split(vertices, faces, name=None, static=False):
    """`vertex_normals` for meshes where every vertex belongs to exactly one face (dirt/lighting.py:97-129):
    each vertex gets its face's unit normal (not renormalised; vertices used by no face get zero)."""
    vertices, faces = _prepare_vertices_and_faces(vertices, faces)
    vertices = vertices[..., :3]
    normals_by_face = _get_face_normals(vertices, faces)  # [*, F, 3]
    out = torch.zeros_like(vertices)
    for k in range(3):
        out = out.index_add(-2, faces[:, k], normals_by_face)  # scatter_nd sums duplicates, as tf.scatter_nd does
    return out


def split_vertices_by_face(vertices, faces, name=None):
    """An equivalent mesh in which each vertex is used by exactly one face (dirt/lighting.py:132-172):
    returns (new_vertices [*, 3F, 3|4], new_faces [F, 3] = arange(3F))."""
    vertices, faces = _prepare_vertices_and_faces(vertices, faces)
    new_vertices = vertices[..., faces.reshape(-1), :]
    new_faces = torch.arange(faces.shape[0] * 3, dtype=torch.int32, device=vertices.device).reshape(-1, 3)
    return new_vertices, new_faces


def _as(x, like):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=like.dtype, device=like.device)


def _clamp_cosines(cosines, double_sided):
    return torch.abs(cosines) if double_sided else torch.clamp(cosines, min=0.)


def diffuse_directional(vertex_normals, vertex_colors, light_direction, light_color, double_sided=True, name=None):
    """Lambertian reflectance under one directional light (dirt/lighting.py:175-218).  Normals and the light
    direction are assumed normalised.  [*, V, 3], [*, V, C], [*, 3], [*, C] -> [*, V, C]."""
    vertex_normals = _as(vertex_normals, torch.zeros((), dtype=torch.float32))
    vertex_colors = _as(vertex_colors, vertex_normals)
    light_direction = _as(light_direction, vertex_normals)
    light_color = _as(light_color, vertex_normals)
    cosines = torch.matmul(vertex_normals, -light_direction[..., None])  # [*, V, 1]
    cosines = _clamp_cosines(cosines, double_sided)
    return light_color[..., None, :] * vertex_colors * cosines


def specular_directional(vertex_positions, vertex_normals, vertex_reflectivities, light_direction, light_color,
                         camera_position, shininess, double_sided=True, name=None):
    """Phong reflectance under one directional light (dirt/lighting.py:221-283), including the reference's
    placement of the 1e-12 (added to the normalised view vector, dirt/lighting.py:274)."""
    vertex_positions = _as(vertex_positions, torch.zeros((), dtype=torch.float32))
    vertex_normals = _as(vertex_normals, vertex_positions)
    vertex_reflectivities = _as(vertex_reflectivities, vertex_positions)
    light_direction = _as(light_direction, vertex_positions)
    light_color = _as(light_color, vertex_positions)
    camera_position = _as(camera_position, vertex_positions)
    shininess = _as(shininess, vertex_positions)
    to_light = -light_direction
    reflected = -to_light[..., None, :] + 2. * torch.matmul(vertex_normals, to_light[..., None]) * vertex_normals
    to_camera = camera_position[..., None, :] - vertex_positions
    cosines = torch.sum((to_camera / torch.linalg.vector_norm(to_camera, dim=-1, keepdim=True) + 1.e-12) * reflected,
                        dim=-1, keepdim=True)
    cosines = _clamp_cosines(cosines, double_sided)
    return light_color[..., None, :] * vertex_reflectivities * torch.pow(cosines, shininess[..., None, None])


def diffuse_point(vertex_positions, vertex_normals, vertex_colors, light_position, light_color, double_sided=True,
                  name=None):
    """Lambertian reflectance under one point light (dirt/lighting.py:286-344).  As in the reference the cosine
    is taken between the normal and the direction FROM the light TO the point."""
    vertex_positions = _as(vertex_positions, torch.zeros((), dtype=torch.float32))
    vertex_normals = _as(vertex_normals, vertex_positions)
    vertex_colors = _as(vertex_colors, vertex_positions)
    light_position = _as(light_position, vertex_positions)
    light_color = _as(light_color, vertex_positions)
    relative = vertex_positions - light_position[..., None, :]
    incident = relative / (torch.linalg.vector_norm(relative, dim=-1, keepdim=True) + 1.e-12)
    cosines = _clamp_cosines(torch.sum(vertex_normals * incident, dim=-1), double_sided)
    return light_color[..., None, :] * vertex_colors * cosines[..., None]
