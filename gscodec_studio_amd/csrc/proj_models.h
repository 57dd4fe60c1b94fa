// proj_models.h -- camera models shared by the fused projection kernels (projection.hip) and the
// unfused public ops (unfused.hip): Jacobians of pinhole / orthographic / equidistant-fisheye
// projection at a camera-space point and the matching mean VJP.
// Reference: gsplat/cuda/include/proj.cuh (persp 80-199, ortho 9-77, fisheye 202-343).
#pragma once

#include "gs_common.h"

struct Camera {
    Mat3 W;       // world->camera rotation
    float tx, ty, tz;
    float fx, fy, cx, cy;
};

GS_DEV Camera load_camera(const float *__restrict__ viewmats, const float *__restrict__ Ks, uint32_t c) {
    const float *V = viewmats + 16 * c;
    const float *K = Ks + 9 * c;
    Camera cam;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) cam.W.m[i][j] = V[4 * i + j];
    cam.tx = V[3]; cam.ty = V[7]; cam.tz = V[11];
    cam.fx = K[0]; cam.cx = K[2]; cam.fy = K[4]; cam.cy = K[5];
    return cam;
}

// 2x3 Jacobian of the camera model at pc, plus the projected mean.
struct Jac {
    float j00, j01, j02, j10, j11, j12;
};

// pinhole: gsplat/cuda/include/proj.cuh:80-119 (the x/z, y/z clamp only affects J)
GS_DEV void pinhole_jac(const Camera &cam, float x, float y, float z, int W, int H,
                        Jac &J, float &mx, float &my, float &txc, float &tyc) {
    float tan_fovx = 0.5f * W / cam.fx;
    float tan_fovy = 0.5f * H / cam.fy;
    float lim_x_pos = (W - cam.cx) / cam.fx + 0.3f * tan_fovx;
    float lim_x_neg = cam.cx / cam.fx + 0.3f * tan_fovx;
    float lim_y_pos = (H - cam.cy) / cam.fy + 0.3f * tan_fovy;
    float lim_y_neg = cam.cy / cam.fy + 0.3f * tan_fovy;
    float rz = 1.f / z;
    float rz2 = rz * rz;
    txc = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, x * rz));
    tyc = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, y * rz));
    J.j00 = cam.fx * rz; J.j01 = 0.f;         J.j02 = -cam.fx * txc * rz2;
    J.j10 = 0.f;         J.j11 = cam.fy * rz; J.j12 = -cam.fy * tyc * rz2;
    mx = cam.fx * x * rz + cam.cx;
    my = cam.fy * y * rz + cam.cy;
}

// orthographic: proj.cuh:9-37
GS_DEV void ortho_jac(const Camera &cam, float x, float y, Jac &J, float &mx, float &my) {
    J.j00 = cam.fx; J.j01 = 0.f; J.j02 = 0.f;
    J.j10 = 0.f; J.j11 = cam.fy; J.j12 = 0.f;
    mx = cam.fx * x + cam.cx;
    my = cam.fy * y + cam.cy;
}

// fisheye (equidistant): proj.cuh:202-243
struct FisheyeTerms {
    float x2, y2, xy, r2, rho, inv_rho, len, theta, a, b;
};

GS_DEV FisheyeTerms fisheye_terms(float x, float y, float z) {
    const float eps = 0.0000001f;
    FisheyeTerms t;
    t.x2 = x * x + eps;
    t.y2 = y * y;
    t.xy = x * y;
    t.r2 = t.x2 + t.y2;
    t.rho = t.r2 + z * z;
    t.inv_rho = 1.f / t.rho;
    t.len = sqrtf(x * x + y * y) + eps;
    t.theta = atan2f(t.len, z);
    t.b = t.theta / t.len / t.r2;
    t.a = z * t.inv_rho / t.r2;
    return t;
}

GS_DEV void fisheye_jac(const Camera &cam, float x, float y, float z, Jac &J, float &mx, float &my) {
    const float eps = 0.0000001f;
    FisheyeTerms t = fisheye_terms(x, y, z);
    float theta_m = atan2f(t.len, z + eps);
    mx = x * cam.fx * theta_m / t.len + cam.cx;
    my = y * cam.fy * theta_m / t.len + cam.cy;
    J.j00 = cam.fx * (t.x2 * t.a + t.y2 * t.b);
    J.j01 = cam.fx * t.xy * (t.a - t.b);
    J.j02 = -cam.fx * x * t.inv_rho;
    J.j10 = cam.fy * t.xy * (t.a - t.b);
    J.j11 = cam.fy * (t.y2 * t.a + t.x2 * t.b);
    J.j12 = -cam.fy * y * t.inv_rho;
}


// d loss / d pc from (v_mean2d, v_J): the camera-model specific tail of the projection VJP
// (pinhole proj.cuh:164-199 incl. the fov-clamp branches, ortho 62-77, fisheye 245-343).
GS_DEV void proj_mean_vjp(const Camera &cam, int camera_model, float x, float y, float z, int W, int H,
                          float v_mx, float v_my, float vj00, float vj01, float vj02, float vj10, float vj11, float vj12,
                          float &vx, float &vy, float &vz) {
    float txc = 0.f, tyc = 0.f;
    FisheyeTerms ft = {};
    Jac J = {};
    if (camera_model == GS_CAMERA_PINHOLE) {
        float mx, my;
        pinhole_jac(cam, x, y, z, W, H, J, mx, my, txc, tyc);
    } else if (camera_model == GS_CAMERA_FISHEYE) {
        float mx, my;
        fisheye_jac(cam, x, y, z, J, mx, my);
        ft = fisheye_terms(x, y, z);
    }
    if (camera_model == GS_CAMERA_PINHOLE) {
        // proj.cuh:122-199
        float rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz;
        float tan_fovx = 0.5f * W / cam.fx, tan_fovy = 0.5f * H / cam.fy;
        float lim_x_pos = (W - cam.cx) / cam.fx + 0.3f * tan_fovx;
        float lim_x_neg = cam.cx / cam.fx + 0.3f * tan_fovx;
        float lim_y_pos = (H - cam.cy) / cam.fy + 0.3f * tan_fovy;
        float lim_y_neg = cam.cy / cam.fy + 0.3f * tan_fovy;
        vx = cam.fx * rz * v_mx;
        vy = cam.fy * rz * v_my;
        vz = -(cam.fx * x * v_mx + cam.fy * y * v_my) * rz2;
        float xr = x * rz, yr = y * rz;
        if (xr <= lim_x_pos && xr >= -lim_x_neg) vx += -cam.fx * rz2 * vj02;
        else vz += -cam.fx * rz3 * vj02 * txc;
        if (yr <= lim_y_pos && yr >= -lim_y_neg) vy += -cam.fy * rz2 * vj12;
        else vz += -cam.fy * rz3 * vj12 * tyc;
        vz += -cam.fx * rz2 * vj00 - cam.fy * rz2 * vj11 + 2.f * cam.fx * txc * rz3 * vj02 +
              2.f * cam.fy * tyc * rz3 * vj12;
    } else if (camera_model == GS_CAMERA_ORTHO) {
        vx = cam.fx * v_mx;
        vy = cam.fy * v_my;
        vz = 0.f;
    } else {
        // fisheye: mean2d gradient through J itself (J is the Jacobian of the map), plus d J / d pc.
        // With r = |(x, y)|, rho = r^2 + z^2, theta = atan2(r, z):
        //     J = [ fx (s + x^2 u),  fx x y u,  -fx x / rho ;  fy x y u,  fy (s + y^2 u),  -fy y / rho ]
        //     s = theta / r,   u = (z / rho - s) / r^2,   w = (-2 z / rho^2 - 3 u) / r^2
        // and  ds = (u x, u y, -1/rho),  du = (w x, w y, 2/rho^2).  Equivalent in exact arithmetic to
        // proj.cuh:245-343, but u and w are differences of nearly equal terms near the optical axis
        // (relative size (r/z)^2), so there they come from their series in (r/z)^2 -- the direct
        // form lost 3 digits at r/z ~ 0.04 (measured against float64).
        vx = J.j00 * v_mx + J.j10 * v_my;
        vy = J.j01 * v_mx + J.j11 * v_my;
        vz = J.j02 * v_mx + J.j12 * v_my;
        const float r2 = ft.r2, inv_rho = ft.inv_rho, inv_rho2 = inv_rho * inv_rho;
        const float s_ = ft.theta / ft.len;
        float u, w;
        const float t2 = r2 / (z * z);
        if (z > 0.f && t2 < 0.04f) {
            const float iz = 1.f / z, iz3 = iz * iz * iz;
            u = iz3 * (-2.f / 3.f + t2 * (4.f / 5.f + t2 * (-6.f / 7.f + t2 * (8.f / 9.f - t2 * (10.f / 11.f)))));
            w = iz3 * iz * iz * (8.f / 5.f + t2 * (-24.f / 7.f + t2 * (16.f / 3.f + t2 * (-80.f / 11.f + t2 * (120.f / 13.f)))));
        } else {
            u = (z * inv_rho - s_) / r2;
            w = (-2.f * z * inv_rho2 - 3.f * u) / r2;
        }
        const float x2 = x * x, y2 = y * y, xy = x * y;
        const float d00x = 3.f * x * u + x2 * x * w, d00y = y * (u + x2 * w), d00z = -inv_rho + 2.f * x2 * inv_rho2;
        const float d01x = y * (u + x2 * w), d01y = x * (u + y2 * w), d01z = 2.f * xy * inv_rho2;
        const float d02x = -inv_rho + 2.f * x2 * inv_rho2, d02y = 2.f * xy * inv_rho2, d02z = 2.f * x * z * inv_rho2;
        const float d11x = x * (u + y2 * w), d11y = 3.f * y * u + y2 * y * w, d11z = -inv_rho + 2.f * y2 * inv_rho2;
        const float d12x = 2.f * xy * inv_rho2, d12y = -inv_rho + 2.f * y2 * inv_rho2, d12z = 2.f * y * z * inv_rho2;
        vx += cam.fx * (d00x * vj00 + d01x * vj01 + d02x * vj02) + cam.fy * (d01x * vj10 + d11x * vj11 + d12x * vj12);
        vy += cam.fx * (d00y * vj00 + d01y * vj01 + d02y * vj02) + cam.fy * (d01y * vj10 + d11y * vj11 + d12y * vj12);
        vz += cam.fx * (d00z * vj00 + d01z * vj01 + d02z * vj02) + cam.fy * (d01z * vj10 + d11z * vj11 + d12z * vj12);
    }
}
