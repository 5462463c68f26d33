"""TEST INFRASTRUCTURE -- a torch restatement of the two stages of the path the reference runs in PyTorch, for bench.py's
`cpu_baseline` leg (CPU baseline (1)): the reference has no CPU rasteriser, but `project_gaussians`
(gs/renderer.py:366-421, with utils/transforms.py:34-46 and kornia 0.6.0's quaternion_to_rotation_matrix) and
`tile_culling_aabb_count` (gs/culling.py:8-37, utils/camera.py:301-314) are plain torch and run on the CPU as written.
/root/reference does not exist on the GPU box, so they are restated here line for line (kind "port");
tests/test_oracle_golden.py::test_torch_port_is_the_references_torch_code holds this file to the reference's own functions,
imported, wherever /root/reference is present.  Nothing in the product imports it."""
import torch


def quaternion_to_rotation_matrix_wxyz(q):
    """kornia 0.6.0 geometry/conversions.py quaternion_to_rotation_matrix(order=WXYZ): normalise, then the standard matrix"""
    q = torch.nn.functional.normalize(q, p=2.0, dim=-1, eps=1e-12)
    w, x, y, z = torch.chunk(q, chunks=4, dim=-1)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.tensor(1.0)
    m = torch.stack((one - (tyy + tzz), txy - twz, txz + twy, txy + twz, one - (txx + tzz), tyz - twx, txz - twy, tyz + twx,
                     one - (txx + tyy)), dim=-1).view(-1, 3, 3)
    return m


@torch.no_grad()
def jacobian(u):  # gs/renderer.py:366-378
    l = torch.norm(u, dim=-1)
    J = torch.zeros(u.size(0), 3, 3).to(u)
    J[..., 0, 0] = 1.0 / u[..., 2]
    J[..., 2, 0] = u[..., 0] / l
    J[..., 1, 1] = 1.0 / u[..., 2]
    J[..., 2, 1] = u[..., 1] / l
    J[..., 0, 2] = -u[..., 0] / u[..., 2] / u[..., 2]
    J[..., 1, 2] = -u[..., 1] / u[..., 2] / u[..., 2]
    J[..., 2, 2] = u[..., 2] / l
    return J


def project_gaussians(mean, qvec, svec, c2w, detach_depth=False):  # gs/renderer.py:381-421
    d = -c2w[..., :3, 3]
    W = torch.transpose(c2w[..., :3, :3], -1, -2)
    projected_mean = torch.einsum("ij,bj->bi", W, mean + d)
    rotmat = svec.unsqueeze(-2) * quaternion_to_rotation_matrix_wxyz(qvec)  # utils/transforms.py:34-46
    sigma = rotmat @ torch.transpose(rotmat, -1, -2)
    J = jacobian(projected_mean)
    JW = torch.einsum("bij,jk->bik", J, W)
    projected_cov = torch.bmm(torch.bmm(JW, sigma), torch.transpose(JW, -1, -2))[..., :2, :2].contiguous()
    depth = projected_mean[..., 2:].clone().contiguous()
    if detach_depth:
        projected_mean = projected_mean[..., :2].contiguous() / depth.detach()
    else:
        projected_mean = projected_mean[..., :2].contiguous() / depth
    return projected_mean, projected_cov, JW, depth


@torch.no_grad()
def tile_culling_aabb_count(mean, cov, tile_size, fx, fy, cx, cy, w, h, D):  # gs/culling.py:8-37
    aabb_sidelength = torch.stack([torch.sqrt(D * cov[:, 0, 0]), torch.sqrt(D * cov[:, 1, 1])], dim=-1)

    def to_pixels(pts):  # utils/camera.py:301-314
        pts = pts.clone()
        pts[:, 0] = pts[:, 0] * fx + cx
        pts[:, 1] = pts[:, 1] * fy + cy
        return pts.to(torch.int32)
    topleft_pixels, bottomright_pixels = to_pixels(mean - aabb_sidelength), to_pixels(mean + aabb_sidelength)
    topleft_pixels[..., 0].clamp_(min=0, max=w - 1)
    topleft_pixels[..., 1].clamp_(min=0, max=h - 1)
    bottomright_pixels[..., 0].clamp_(min=0, max=w - 1)
    bottomright_pixels[..., 1].clamp_(min=0, max=h - 1)
    topleft_pixels = torch.div(topleft_pixels, tile_size, rounding_mode="floor")
    bottomright_pixels = torch.div(bottomright_pixels, tile_size, rounding_mode="floor")
    N_with_dub = torch.prod(bottomright_pixels - topleft_pixels + 1, dim=-1).sum().item()
    return N_with_dub, topleft_pixels, bottomright_pixels
