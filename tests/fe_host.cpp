// Host build of the device field arithmetic (kangaroo_amd/csrc/fe29.h, modinv30.h) for CPU-side
// unit tests.  Test-only: the product runs these functions inside HIP kernels exclusively.
#include "../kangaroo_amd/csrc/fe29.h"
#include "../kangaroo_amd/csrc/modinv30.h"
using namespace kg;

extern "C" {
// all values: 8 little-endian u32 words
void fh_mul(u32 r[8], const u32 a[8], const u32 b[8]) {
  fe x, y, z; fe_from_words(x, a); fe_from_words(y, b); fe_mul(z, x, y); fe_canon(z); fe_to_words(r, z);
}
void fh_sqr(u32 r[8], const u32 a[8]) {
  fe x, z; fe_from_words(x, a); fe_sqr(z, x); fe_canon(z); fe_to_words(r, z);
}
void fh_sub(u32 r[8], const u32 a[8], const u32 b[8]) {
  fe x, y, z; fe_from_words(x, a); fe_from_words(y, b); fe_sub_wide(z, x, y); fe_canon(z); fe_to_words(r, z);
}
void fh_sub2(u32 r[8], const u32 a[8], const u32 b[8], const u32 c[8]) {
  fe x, y, w, z; fe_from_words(x, a); fe_from_words(y, b); fe_from_words(w, c);
  fe_sub2_wide(z, x, y, w); fe_canon(z); fe_to_words(r, z);
}
// (a - b) * c : exercises wide x weak multiplication
void fh_submul(u32 r[8], const u32 a[8], const u32 b[8], const u32 c[8]) {
  fe x, y, w, d, z; fe_from_words(x, a); fe_from_words(y, b); fe_from_words(w, c);
  fe_sub_wide(d, x, y); fe_mul(z, d, w); fe_canon(z); fe_to_words(r, z);
}
void fh_canon(u32 r[8], const u32 a[8]) { fe x; fe_from_words(x, a); fe_canon(x); fe_to_words(r, x); }
void fh_inv(u32 r[8], const u32 a[8]) { fe x, z; fe_from_words(x, a); fe_inv_t<0>(z, x); fe_canon(z); fe_to_words(r, z); }
void fh_inv_var(u32 r[8], const u32 a[8]) { fe x, z; fe_from_words(x, a); fe_inv_t<1>(z, x); fe_canon(z); fe_to_words(r, z); }
void fh_inv_fix(u32 r[8], const u32 a[8]) { fe x, z; fe_from_words(x, a); fe_inv_t<2>(z, x); fe_canon(z); fe_to_words(r, z); }
// one affine jump exactly as the kernel's back pass does it, given dx^-1
void fh_jump(u32 rx[8], u32 ry[8], const u32 x[8], const u32 y[8], const u32 jx[8], const u32 jy[8]) {
  fe X, Y, JX, JY, dx, dxi, dy, s, p2, RX, t, RY;
  fe_from_words(X, x); fe_from_words(Y, y); fe_from_words(JX, jx); fe_from_words(JY, jy);
  fe_sub_wide(dx, X, JX); fe_inv(dxi, dx);
  fe_sub_wide(dy, Y, JY); fe_mul(s, dy, dxi); fe_sqr(p2, s);
  fe_sub2_wide(RX, p2, JX, X); fe_canon(RX);
  fe_sub_wide(t, X, RX); fe_mul(RY, t, s); fe_sub_wide(RY, RY, Y); fe_canon(RY);
  fe_to_words(rx, RX); fe_to_words(ry, RY);
}
}
