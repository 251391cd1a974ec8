// glm-ordered matrix constructors and inverses used by the host facade.
#include "hmath.h"

namespace pth {

Mat4 Translate(const Mat4 &M, Vec3 v) {
    Mat4 R = M;
    for (int r = 0; r < 4; ++r)
        R.at(3, r) = ((M.at(0, r) * v.x + M.at(1, r) * v.y) + M.at(2, r) * v.z) + M.at(3, r);
    return R;
}

Mat4 Scale(const Mat4 &M, Vec3 v) {
    Mat4 R = M;
    for (int r = 0; r < 4; ++r) {
        R.at(0, r) = M.at(0, r) * v.x;
        R.at(1, r) = M.at(1, r) * v.y;
        R.at(2, r) = M.at(2, r) * v.z;
    }
    return R;
}

Mat4 Rotate(const Mat4 &M, float angle, Vec3 v) {
    const float c = std::cos(angle), s = std::sin(angle);
    const Vec3 axis = Normalize(v);
    const Vec3 temp = axis * (1.0f - c);
    float rot[3][3];
    rot[0][0] = c + temp.x * axis.x;
    rot[0][1] = temp.x * axis.y + s * axis.z;
    rot[0][2] = temp.x * axis.z - s * axis.y;
    rot[1][0] = temp.y * axis.x - s * axis.z;
    rot[1][1] = c + temp.y * axis.y;
    rot[1][2] = temp.y * axis.z + s * axis.x;
    rot[2][0] = temp.z * axis.x + s * axis.y;
    rot[2][1] = temp.z * axis.y - s * axis.x;
    rot[2][2] = c + temp.z * axis.z;
    Mat4 R;
    for (int j = 0; j < 3; ++j)
        for (int r = 0; r < 4; ++r)
            R.at(j, r) = (M.at(0, r) * rot[j][0] + M.at(1, r) * rot[j][1]) + M.at(2, r) * rot[j][2];
    for (int r = 0; r < 4; ++r) R.at(3, r) = M.at(3, r);
    return R;
}

Mat4 FromQuat(float x, float y, float z, float w) {
    Mat4 R = Mat4::Identity();
    const float qxx = x * x, qyy = y * y, qzz = z * z;
    const float qxz = x * z, qxy = x * y, qyz = y * z;
    const float qwx = w * x, qwy = w * y, qwz = w * z;
    R.at(0, 0) = 1.0f - 2.0f * (qyy + qzz);
    R.at(0, 1) = 2.0f * (qxy + qwz);
    R.at(0, 2) = 2.0f * (qxz - qwy);
    R.at(1, 0) = 2.0f * (qxy - qwz);
    R.at(1, 1) = 1.0f - 2.0f * (qxx + qzz);
    R.at(1, 2) = 2.0f * (qyz + qwx);
    R.at(2, 0) = 2.0f * (qxz + qwy);
    R.at(2, 1) = 2.0f * (qyz - qwx);
    R.at(2, 2) = 1.0f - 2.0f * (qxx + qyy);
    return R;
}

// 2x2 sub-determinant of rows (r0,r1) taken from columns (c0,c1).
static inline float Minor2(const Mat4 &M, int c0, int r0, int c1, int r1) {
    return M.at(c0, r0) * M.at(c1, r1) - M.at(c1, r0) * M.at(c0, r1);
}

Mat4 Inverse(const Mat4 &M) {
    // cofactor expansion in glm's grouping: six 2x2 factors per column pair, then signed 4-vectors
    const float c00 = Minor2(M, 2, 2, 3, 3), c02 = Minor2(M, 1, 2, 3, 3), c03 = Minor2(M, 1, 2, 2, 3);
    const float c04 = Minor2(M, 2, 1, 3, 3), c06 = Minor2(M, 1, 1, 3, 3), c07 = Minor2(M, 1, 1, 2, 3);
    const float c08 = Minor2(M, 2, 1, 3, 2), c10 = Minor2(M, 1, 1, 3, 2), c11 = Minor2(M, 1, 1, 2, 2);
    const float c12 = Minor2(M, 2, 0, 3, 3), c14 = Minor2(M, 1, 0, 3, 3), c15 = Minor2(M, 1, 0, 2, 3);
    const float c16 = Minor2(M, 2, 0, 3, 2), c18 = Minor2(M, 1, 0, 3, 2), c19 = Minor2(M, 1, 0, 2, 2);
    const float c20 = Minor2(M, 2, 0, 3, 1), c22 = Minor2(M, 1, 0, 3, 1), c23 = Minor2(M, 1, 0, 2, 1);
    const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    const float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {M.at(1, 0), M.at(0, 0), M.at(0, 0), M.at(0, 0)};
    const float v1[4] = {M.at(1, 1), M.at(0, 1), M.at(0, 1), M.at(0, 1)};
    const float v2[4] = {M.at(1, 2), M.at(0, 2), M.at(0, 2), M.at(0, 2)};
    const float v3[4] = {M.at(1, 3), M.at(0, 3), M.at(0, 3), M.at(0, 3)};
    Mat4 R;
    for (int k = 0; k < 4; ++k) {
        const float sa = (k & 1) ? -1.0f : 1.0f, sb = -sa;
        R.at(0, k) = ((v1[k] * f0[k] - v2[k] * f1[k]) + v3[k] * f2[k]) * sa;
        R.at(1, k) = ((v0[k] * f0[k] - v2[k] * f3[k]) + v3[k] * f4[k]) * sb;
        R.at(2, k) = ((v0[k] * f1[k] - v1[k] * f3[k]) + v3[k] * f5[k]) * sa;
        R.at(3, k) = ((v0[k] * f2[k] - v1[k] * f4[k]) + v2[k] * f5[k]) * sb;
    }
    const float d0 = M.at(0, 0) * R.at(0, 0), d1 = M.at(0, 1) * R.at(1, 0);
    const float d2 = M.at(0, 2) * R.at(2, 0), d3 = M.at(0, 3) * R.at(3, 0);
    const float inv_det = 1.0f / ((d0 + d1) + (d2 + d3));
    for (float &e : R.m) e = e * inv_det;
    return R;
}

Mat4 InverseTranspose(const Mat4 &M) {
    const float s00 = Minor2(M, 2, 2, 3, 3), s01 = Minor2(M, 2, 1, 3, 3), s02 = Minor2(M, 2, 1, 3, 2);
    const float s03 = Minor2(M, 2, 0, 3, 3), s04 = Minor2(M, 2, 0, 3, 2), s05 = Minor2(M, 2, 0, 3, 1);
    const float s06 = Minor2(M, 1, 2, 3, 3), s07 = Minor2(M, 1, 1, 3, 3), s08 = Minor2(M, 1, 1, 3, 2);
    const float s09 = Minor2(M, 1, 0, 3, 3), s10 = Minor2(M, 1, 0, 3, 2), s11 = Minor2(M, 1, 0, 3, 1);
    const float s12 = Minor2(M, 1, 2, 2, 3), s13 = Minor2(M, 1, 1, 2, 3), s14 = Minor2(M, 1, 1, 2, 2);
    const float s15 = Minor2(M, 1, 0, 2, 3), s16 = Minor2(M, 1, 0, 2, 2), s17 = Minor2(M, 1, 0, 2, 1);
    Mat4 I;
    I.at(0, 0) = +((M.at(1, 1) * s00 - M.at(1, 2) * s01) + M.at(1, 3) * s02);
    I.at(0, 1) = -((M.at(1, 0) * s00 - M.at(1, 2) * s03) + M.at(1, 3) * s04);
    I.at(0, 2) = +((M.at(1, 0) * s01 - M.at(1, 1) * s03) + M.at(1, 3) * s05);
    I.at(0, 3) = -((M.at(1, 0) * s02 - M.at(1, 1) * s04) + M.at(1, 2) * s05);
    I.at(1, 0) = -((M.at(0, 1) * s00 - M.at(0, 2) * s01) + M.at(0, 3) * s02);
    I.at(1, 1) = +((M.at(0, 0) * s00 - M.at(0, 2) * s03) + M.at(0, 3) * s04);
    I.at(1, 2) = -((M.at(0, 0) * s01 - M.at(0, 1) * s03) + M.at(0, 3) * s05);
    I.at(1, 3) = +((M.at(0, 0) * s02 - M.at(0, 1) * s04) + M.at(0, 2) * s05);
    I.at(2, 0) = +((M.at(0, 1) * s06 - M.at(0, 2) * s07) + M.at(0, 3) * s08);
    I.at(2, 1) = -((M.at(0, 0) * s06 - M.at(0, 2) * s09) + M.at(0, 3) * s10);
    I.at(2, 2) = +((M.at(0, 0) * s07 - M.at(0, 1) * s09) + M.at(0, 3) * s11);
    I.at(2, 3) = -((M.at(0, 0) * s08 - M.at(0, 1) * s10) + M.at(0, 2) * s11);
    I.at(3, 0) = -((M.at(0, 1) * s12 - M.at(0, 2) * s13) + M.at(0, 3) * s14);
    I.at(3, 1) = +((M.at(0, 0) * s12 - M.at(0, 2) * s15) + M.at(0, 3) * s16);
    I.at(3, 2) = -((M.at(0, 0) * s13 - M.at(0, 1) * s15) + M.at(0, 3) * s17);
    I.at(3, 3) = +((M.at(0, 0) * s14 - M.at(0, 1) * s16) + M.at(0, 2) * s17);
    const float det = ((M.at(0, 0) * I.at(0, 0) + M.at(0, 1) * I.at(0, 1)) + M.at(0, 2) * I.at(0, 2)) +
                      M.at(0, 3) * I.at(0, 3);
    for (float &e : I.m) e = e / det;
    return I;
}

Mat4 GetTransformMatrix(Vec3 translate, Vec3 rotate_deg, Vec3 scale) {
    const Mat4 I = Mat4::Identity();
    const Mat4 S = Scale(I, scale);
    Mat4 R = Rotate(I, Radians(rotate_deg.x), {1, 0, 0});
    R = Rotate(R, Radians(rotate_deg.y), {0, 1, 0});
    R = Rotate(R, Radians(rotate_deg.z), {0, 0, 1});
    const Mat4 T = Translate(I, translate);
    return Mul(Mul(T, R), S);
}

}  // namespace pth
