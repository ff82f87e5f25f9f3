// tfhe_oracle_kat.c — TEST INFRASTRUCTURE: the reference's deterministic randomness, restated so
// tests/kat_vectors.py can regenerate apps/test-vectors' golden files at full size.
//
//   * tfhe-csprng software generator: AES-128, counter mode over a linear byte table; block t =
//     AES_key(t as little-endian u128), key = seed as little-endian u128
//     (tfhe-csprng/src/generators/aes_ctr/{generic.rs:84-107,states.rs:87-122},
//      implem/soft/block_cipher.rs:27-40,70-80).
//   * Gaussian torus noise: Marsaglia polar method on two i64 draws scaled by 2^-63, first output
//     only, then FromTorus<f64> for u64
//     (tfhe/src/core_crypto/commons/math/random/gaussian.rs:40-69,151-163; math/torus/mod.rs:72-78).
//     ln/sqrt are the C library's (Rust's f64::ln/sqrt lower to the same libm calls).
#include <math.h>
#include <stdint.h>
#include <string.h>

static uint8_t SBOX[256];
static int sbox_ready = 0;

static uint8_t xtime(uint8_t a) { return (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0)); }

static uint8_t gf_mul(uint8_t a, uint8_t b) {
  uint8_t r = 0;
  while (b) {
    if (b & 1) r ^= a;
    a = xtime(a);
    b >>= 1;
  }
  return r;
}

static void sbox_init(void) {
  if (sbox_ready) return;
  // multiplicative inverse by search (256*256, once), then the FIPS-197 affine map
  for (int x = 0; x < 256; ++x) {
    uint8_t inv = 0;
    if (x)
      for (int y = 1; y < 256; ++y)
        if (gf_mul((uint8_t)x, (uint8_t)y) == 1) {
          inv = (uint8_t)y;
          break;
        }
    uint8_t s = inv, r = inv;
    for (int i = 0; i < 4; ++i) {
      s = (uint8_t)((s << 1) | (s >> 7));
      r ^= s;
    }
    SBOX[x] = r ^ 0x63;
  }
  sbox_ready = 1;
}

typedef struct {
  uint8_t rk[11][16];
} aes128_t;

static void aes128_init(aes128_t *a, const uint8_t key[16]) {
  sbox_init();
  memcpy(a->rk[0], key, 16);
  uint8_t rcon = 1;
  for (int r = 1; r <= 10; ++r) {
    const uint8_t *p = a->rk[r - 1];
    uint8_t *o = a->rk[r];
    uint8_t t[4] = {SBOX[p[13]], SBOX[p[14]], SBOX[p[15]], SBOX[p[12]]};
    t[0] ^= rcon;
    rcon = xtime(rcon);
    for (int i = 0; i < 4; ++i) o[i] = p[i] ^ t[i];
    for (int i = 4; i < 16; ++i) o[i] = p[i] ^ o[i - 4];
  }
}

static void aes128_encrypt(const aes128_t *a, const uint8_t in[16], uint8_t out[16]) {
  uint8_t s[16], t[16];
  for (int i = 0; i < 16; ++i) s[i] = in[i] ^ a->rk[0][i];
  for (int r = 1; r <= 10; ++r) {
    // SubBytes + ShiftRows (column-major state: byte index = 4*col + row)
    for (int c = 0; c < 4; ++c)
      for (int row = 0; row < 4; ++row) t[4 * c + row] = SBOX[s[4 * ((c + row) & 3) + row]];
    if (r < 10) {
      for (int c = 0; c < 4; ++c) {
        const uint8_t *q = t + 4 * c;
        const uint8_t all = q[0] ^ q[1] ^ q[2] ^ q[3];
        s[4 * c + 0] = q[0] ^ all ^ xtime(q[0] ^ q[1]);
        s[4 * c + 1] = q[1] ^ all ^ xtime(q[1] ^ q[2]);
        s[4 * c + 2] = q[2] ^ all ^ xtime(q[2] ^ q[3]);
        s[4 * c + 3] = q[3] ^ all ^ xtime(q[3] ^ q[0]);
      }
    } else {
      memcpy(s, t, 16);
    }
    for (int i = 0; i < 16; ++i) s[i] ^= a->rk[r][i];
  }
  memcpy(out, s, 16);
}

static void ctr_block(const aes128_t *a, uint64_t t, uint8_t out[16]) {
  uint8_t in[16] = {0};
  for (int i = 0; i < 8; ++i) in[i] = (uint8_t)(t >> (8 * i));
  aes128_encrypt(a, in, out);
}

// bytes [offset, offset+n) of the generator's linear byte table
void orc_csprng_bytes(const uint8_t key[16], uint64_t offset, uint64_t n, uint8_t *out) {
  aes128_t a;
  aes128_init(&a, key);
  uint8_t blk[16];
  uint64_t done = 0;
  while (done < n) {
    const uint64_t pos = offset + done;
    ctr_block(&a, pos >> 4, blk);
    const uint64_t in_blk = pos & 15;
    uint64_t take = 16 - in_blk;
    if (take > n - done) take = n - done;
    memcpy(out + done, blk + in_blk, take);
    done += take;
  }
}

// `count` Gaussian torus samples drawn sequentially from the byte table starting at `offset`;
// returns the number of bytes consumed (16 per attempt).
uint64_t orc_csprng_gaussian_u64(const uint8_t key[16], uint64_t offset, double std, uint64_t count, uint64_t *out) {
  aes128_t a;
  aes128_init(&a, key);
  uint64_t pos = offset;
  uint8_t buf[32];
  for (uint64_t i = 0; i < count; ++i) {
    for (;;) {
      // 16 bytes starting at pos (may straddle two blocks)
      ctr_block(&a, pos >> 4, buf);
      if (pos & 15) ctr_block(&a, (pos >> 4) + 1, buf + 16);
      const uint8_t *p = buf + (pos & 15);
      pos += 16;
      int64_t iu, iv;
      memcpy(&iu, p, 8);
      memcpy(&iv, p + 8, 8);
      const double u = (double)iu * 0x1p-63, v = (double)iv * 0x1p-63;
      const double s = u * u + v * v;
      if (s > 0.0 && s < 1.0) {
        const double cst = std * sqrt(-2.0 * log(s) / s);
        const double x = u * cst + 0.0;
        double fract = x - round(x);
        fract *= 0x1p64;
        fract = round(fract);
        out[i] = (uint64_t)(int64_t)fract;
        break;
      }
    }
  }
  return pos - offset;
}
