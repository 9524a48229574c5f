// Static set-up of a plan: the iteration-invariant operators (C, G, H, right-hand-side parts, edge tiles) for
// diagonal noise, a dense noise covariance (with or without flags) and the time-dependent modes.
#include "hpx_chain.h"

namespace {


// ---- setup ------------------------------------------------------------------
// Z[b][j][col]: col<TP: Ni (w d)_t + Ni^1/2 omega_b,t ; TP..: Ni F[:,m] ; TP+MP: Ni
// D[b][j][t] = w_j d[t][j]
__global__ void k_prep(const double* __restrict__ vis, const uint8_t* __restrict__ flags,
                       const double* __restrict__ ninv, const double* __restrict__ fg,
                       const int fg_shared, const double* __restrict__ omega,
                       double* __restrict__ Zre, double* __restrict__ Zim,
                       double* __restrict__ Dre, double* __restrict__ Dim,
                       double* __restrict__ ni_out, const int T, const int N, const int M,
                       const int NP, const int TP, const int MP, const int ncol, const int omega_mod,
                       const int mask_data_only) {
  // mask_data_only: the flags mask the data columns only (dense noise with flags: the matrix blocks are
  // those of the unflagged noise, the mask comes in through the Woodbury correction)
  const int b = blockIdx.y;
  const int tom = omega_mod > 0 ? b % omega_mod : 0;      // per-time units: the draws of "their" time
  const long tot = (long)NP * ncol;
  const double* F = fg + (fg_shared ? 0 : (long)b * N * M * 2);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / ncol), col = (int)(e % ncol);
    double zr = 0.0, zi = 0.0;
    if (j < N) {
      const double w = flags[(long)b * N + j] ? 1.0 : 0.0;
      const double ni = ninv[(long)b * N + j] * ((mask_data_only && col >= TP) ? 1.0 : w);
      if (col < TP) {
        const int t = col;
        double dr = 0.0, di = 0.0;
        if (t < T) {
          const long o = (((long)b * T + t) * N + j) * 2;
          dr = vis[o] * w;
          di = vis[o + 1] * w;
          zr = ninv[(long)b * N + j] * dr;
          zi = ninv[(long)b * N + j] * di;
          if (omega) {
            const double nih = sqrt(ni);
            zr += nih * (omega[((long)(t + tom) * 4 + 2) * N + j] / SQRT2);
            zi += nih * (omega[((long)(t + tom) * 4 + 3) * N + j] / SQRT2);
          }
        }
        Dre[((long)b * NP + j) * TP + t] = dr;
        Dim[((long)b * NP + j) * TP + t] = di;
      } else if (col < TP + MP) {
        const int m = col - TP;
        if (m < M) {
          zr = ni * F[((long)j * M + m) * 2];
          zi = ni * F[((long)j * M + m) * 2 + 1];
        }
      } else if (col == TP + MP) {
        zr = ni;
        ni_out[(long)b * N + j] = ni;
      }
    } else if (col < TP) {
      Dre[((long)b * NP + j) * TP + col] = 0.0;
      Dim[((long)b * NP + j) * TP + col] = 0.0;
    }
    Zre[(long)b * tot + e] = zr;
    Zim[(long)b * tot + e] = zi;
  }
}

// shared omega_a block: Z2[j][t] = (omi + i omj)/sqrt2
__global__ void k_prep_omega(const double* __restrict__ omega, double* __restrict__ Zre,
                             double* __restrict__ Zim, const int T, const int N, const int NP,
                             const int TP) {
  const long tot = (long)NP * TP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / TP), t = (int)(e % TP);
    double zr = 0.0, zi = 0.0;
    if (j < N && t < T) {
      zr = omega[((long)t * 4 + 0) * N + j] / SQRT2;
      zi = omega[((long)t * 4 + 1) * N + j] / SQRT2;
    }
    Zre[e] = zr;
    Zim[e] = zi;
  }
}

// circ[m] = R[(m + N/2) mod N][colC] / sqrt(N)
__global__ void k_circ(const double* __restrict__ Rre, const double* __restrict__ Rim,
                       double* __restrict__ Cre, double* __restrict__ Cim, const int N,
                       const int NP, const int ncol, const int colC) {
  const int b = blockIdx.y;
  const double s = 1.0 / sqrt((double)N);
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < N; m += gridDim.x * blockDim.x) {
    const int x = (m + N / 2) % N;
    const long o = ((long)b * NP + x) * ncol + colC;
    Cre[(long)b * N + m] = Rre[o] * s;
    Cim[(long)b * N + m] = Rim[o] * s;
  }
}

// H = F^H Ni F (M x M), P4 = F^H (Ni d + Ni^1/2 omega_b) (M x TP); Z holds the operands.
__global__ void k_small(const double* __restrict__ fg, const int fg_shared,
                        const double* __restrict__ Zre, const double* __restrict__ Zim,
                        double* __restrict__ Hre, double* __restrict__ Him,
                        double* __restrict__ P4re, double* __restrict__ P4im, const int N,
                        const int M, const int NP, const int TP, const int MP, const int ncol) {
  const int b = blockIdx.x;
  const double* F = fg + (fg_shared ? 0 : (long)b * N * M * 2);
  const double* zr = Zre + (long)b * NP * ncol;
  const double* zi = Zim + (long)b * NP * ncol;
  const int nh = M * M, np4 = M * TP;
  for (int e = threadIdx.x; e < nh + np4; e += blockDim.x) {
    int m, col;
    if (e < nh) { m = e / M; col = TP + e % M; }
    else { m = (e - nh) / TP; col = (e - nh) % TP; }
    double sr = 0.0, si = 0.0;
    for (int j = 0; j < N; ++j) {
      const double fr = F[((long)j * M + m) * 2], fi = -F[((long)j * M + m) * 2 + 1];   // conj
      const double ar = zr[(long)j * ncol + col], ai = zi[(long)j * ncol + col];
      sr += fr * ar - fi * ai;
      si += fr * ai + fi * ar;
    }
    if (e < nh) {
      Hre[(long)b * nh + e] = sr;
      Him[(long)b * nh + e] = si;
    } else {
      P4re[(long)b * np4 + (e - nh)] = sr;
      P4im[(long)b * np4 + (e - nh)] = si;
    }
  }
}

__global__ void k_fg_planar(const double* __restrict__ fg, double* __restrict__ Fre,
                            double* __restrict__ Fim, const long tot) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    Fre[e] = fg[2 * e];
    Fim[e] = fg[2 * e + 1];
  }
}

// ---- dense (Hermitian, non-diagonal) inverse noise covariance -----------------------------------
// (nbl|1, N, N) c128 row-major -> planar [b][NP][NP] zero padded; herm != 0: out[k][x] = conj(in[x][k])
__global__ void k_dense_planar(const double* __restrict__ m, const int shared, double* __restrict__ re,
                               double* __restrict__ im, const int N, const int NP) {
  const int b = blockIdx.y;
  const long tot = (long)NP * NP;
  const double* src = m + (shared ? 0 : (long)b * N * N * 2);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / NP), x = (int)(e % NP);
    double vr = 0.0, vi = 0.0;
    if (k < N && x < N) {
      vr = src[((long)k * N + x) * 2];
      vi = src[((long)k * N + x) * 2 + 1];
    }
    re[(long)b * tot + e] = vr;
    im[(long)b * tot + e] = vi;
  }
}
// planar [b][NP][NP]: out[k][x] = conj(in[x][k])
__global__ void k_conj_transpose(const double* __restrict__ ire, const double* __restrict__ iim,
                                 double* __restrict__ ore, double* __restrict__ oim, const int NP) {
  const int b = blockIdx.y;
  const long tot = (long)NP * NP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / NP), x = (int)(e % NP);
    ore[(long)b * tot + e] = ire[(long)b * tot + (long)x * NP + k];
    oim[(long)b * tot + e] = -iim[(long)b * tot + (long)x * NP + k];
  }
}
// the real diagonal of the planar matrices -> (nbl, N)
__global__ void k_take_diag(const double* __restrict__ re, double* __restrict__ dg, const int N, const int NP) {
  const int b = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += gridDim.x * blockDim.x)
    dg[(long)b * N + k] = re[(long)b * NP * NP + (long)k * NP + k];
}
// Z[b][j][t] += A[b][j][t] for t < TP (both with leading dimension ld_z / ld_a)
__global__ void k_add_block(double* __restrict__ zre, double* __restrict__ zim, const long z_bs, const int ld_z,
                            const double* __restrict__ are, const double* __restrict__ aim, const long a_bs,
                            const int ld_a, const int NP, const int TP) {
  const int b = blockIdx.y;
  const long tot = (long)NP * TP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / TP), t = (int)(e % TP);
    zre[(long)b * z_bs + (long)j * ld_z + t] += are[(long)b * a_bs + (long)j * ld_a + t];
    zim[(long)b * z_bs + (long)j * ld_z + t] += aim[(long)b * a_bs + (long)j * ld_a + t];
  }
}
// omega_b block: O[j][t] = (omk + i oml)/sqrt2, replicated per baseline (the dense product is batched)
__global__ void k_prep_omega_b(const double* __restrict__ omega, double* __restrict__ Ore,
                               double* __restrict__ Oim, const int T, const int N, const int NP, const int TP,
                               const int omega_mod) {
  const int b = blockIdx.y;
  const int tom = omega_mod > 0 ? b % omega_mod : 0;      // per-time units: the draws of "their" time
  const long tot = (long)NP * TP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / TP), t = (int)(e % TP);
    double zr = 0.0, zi = 0.0;
    if (j < N && t < T) {
      zr = omega[((long)(t + tom) * 4 + 2) * N + j] / SQRT2;
      zi = omega[((long)(t + tom) * 4 + 3) * N + j] / SQRT2;
    }
    Ore[(long)b * tot + e] = zr;
    Oim[(long)b * tot + e] = zi;
  }
}

// Edge tiles (hpx_internal.h: E): the iteration-invariant part of rows >= rmin for the columns c < rmin,
// in the factor's tile layout.  Right-hand-side rows hold Q un-conjugated (hpx_edge_init adds P2 / a and
// conjugates, as hpx_gen_entry does); everything else is the entry itself.
__global__ __launch_bounds__(256) void k_build_edge(const hpx_gen_batch B, double* __restrict__ E_all,
                                                    const int npad, const int ld) {
  const int b = blockIdx.y;
  hpx_gen_batch B0 = B;
  B0.has_omega = 0;                                   // the invariant part only
  const hpx_gen G = hpx_gen_for(B0, b);
  const int rmin = B.rmin, nrow = ld - rmin;
  double* E = E_all + (long)b * B.e_bstride;
  const long tot = (long)nrow * rmin;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e / nrow), r = rmin + (int)(e % nrow);
    double vr, vi;
    hpx_gen_entry(G, r, c, npad, vr, vi);
    if (r >= npad) vi = -vi;                          // Q itself: the conjugation happens at the use
    const long o = HPX_EIDX(r, c, rmin);
    E[o] = vr;
    E[o + 16] = vi;
  }
}
// P2T[(t >> 4)][c][t & 15] = P2[c][t]
__global__ void k_p2_tiles(const double* __restrict__ p2re, const double* __restrict__ p2im,
                           double* __restrict__ tre, double* __restrict__ tim, const int NP, const int TP) {
  const long tot = (long)NP * TP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e / TP), t = (int)(e % TP);
    const long q = (((long)(t >> 4) * NP + c) << 4) + (t & 15);
    tre[q] = p2re[e];
    tim[q] = p2im[e];
  }
}
// dense noise with flags (hpx_plan_set_static_dense_flagged): unit vectors of the flagged channels into the
// padded time columns T .. T+f-1 of the operand block, so that Z = Ninv [d | e_j ..] carries the Woodbury
// vectors P = B^H Ninv E through the same transforms as the data (their omega / P2 parts stay zero)
__global__ void k_wb_inject(double* __restrict__ Rre, const int32_t* __restrict__ flist,
                            const int32_t* __restrict__ fcount, const int fmax, const int T, const int NP,
                            const int ncol) {
  const int b = blockIdx.x;
  for (int kf = threadIdx.x; kf < fcount[b]; kf += blockDim.x)
    Rre[((long)b * NP + flist[(long)b * fmax + kf]) * ncol + T + kf] = 1.0;
}
// diag of (nu, N, N) c128 matrices -> (nu, N) f64
__global__ void k_pt_diag(const double* __restrict__ m, double* __restrict__ dg, const int N) {
  const int u = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += gridDim.x * blockDim.x)
    dg[(long)u * N + k] = m[(((long)u * N + k) * N + k) * 2];
}
// flags_any[b][x] = AND_t flags_t[b][t][x];  ninv_any[b][x] = ninv_t[b][0][x]
__global__ void k_pt_reduce(const uint8_t* __restrict__ ft, const double* __restrict__ nt,
                            uint8_t* __restrict__ fany, double* __restrict__ nany, const int T, const int N) {
  const int b = blockIdx.y;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < N; x += gridDim.x * blockDim.x) {
    uint8_t a = 1;
    for (int t = 0; t < T; ++t) a &= (ft[((long)b * T + t) * N + x] ? 1 : 0);
    fany[(long)b * N + x] = a;
    nany[(long)b * N + x] = nt[(long)b * T * N + x];
  }
}
// D[b][x][t] = w_bt[x] vis[b][t][x]  (the masked data of pspec.py:613, per time)
__global__ void k_pt_data(const double* __restrict__ vis, const uint8_t* __restrict__ ft,
                          double* __restrict__ Dre, double* __restrict__ Dim, const int T, const int N,
                          const int NP, const int TP) {
  const int b = blockIdx.y;
  const long tot = (long)N * T;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int t = (int)(e / N), x = (int)(e % N);
    const long o = ((long)b * T + t) * N + x;
    const double w = ft[o] ? 1.0 : 0.0;
    Dre[((long)b * NP + x) * TP + t] = w * vis[2 * o];
    Dim[((long)b * NP + x) * TP + t] = w * vis[2 * o + 1];
  }
}
// child's omega_a block: PT[t][x][0] = P2[x][t], other columns zero
__global__ void k_pt_p2(const double* __restrict__ p2re, const double* __restrict__ p2im,
                        double* __restrict__ ptre, double* __restrict__ ptim, const int T, const int NP,
                        const int TP, const int TPc) {
  const long tot = (long)T * NP * TPc;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % TPc), x = (int)((e / TPc) % NP), t = (int)(e / ((long)TPc * NP));
    ptre[e] = (c == 0) ? p2re[(long)x * TP + t] : 0.0;
    ptim[e] = (c == 0) ? p2im[(long)x * TP + t] : 0.0;
  }
}
// child's P2 by row tile: PTT[t][c][0] = P2[c][t]  (rows 1..15 of the unit's only RHS tile stay zero)
__global__ void k_pt_p2t(const double* __restrict__ p2re, const double* __restrict__ p2im,
                         double* __restrict__ tre, double* __restrict__ tim, const int T, const int NP,
                         const int TP) {
  const long tot = (long)T * NP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int t = (int)(e / NP), c = (int)(e % NP);
    tre[e << 4] = p2re[(long)c * TP + t];
    tim[e << 4] = p2im[(long)c * TP + t];
  }
}

}  // namespace


static int set_static_impl(hpx_plan* p, const double* vis, const uint8_t* flags,
                           const double* ninv, const double* ninv_dense, const double* nih_dense,
                           int noise_shared, const double* fgmodes, int fg_shared,
                           const int32_t* prior_map, const double* xgrid, int nxrows,
                           int prior_shared, int ngrid, const double* omega,
                           const double* fop, int any_flags, void* stream, int wb = 0) {
  HPX_REQUIRE(p && vis && flags && (ninv || ninv_dense) && fop && prior_map, "hpx_plan_set_static: null argument");
  HPX_REQUIRE(p->M == 0 || fgmodes, "hpx_plan_set_static: fgmodes required when M > 0");
  HPX_REQUIRE(nxrows == 0 || (xgrid && ngrid >= 2 && ngrid <= 8192),
              "hpx_plan_set_static: bad prior grid");
  hipStream_t st = (hipStream_t)stream;
  const int nbl = p->nbl, N = p->N, M = p->M, T = p->T, NP = p->NP, TP = p->TP, MP = p->MP;
  p->fg_shared = fg_shared ? 1 : 0;
  p->prior_shared = prior_shared ? 1 : 0;
  p->has_omega = omega ? 1 : 0;
  p->any_flags = any_flags ? 1 : 0;
  p->ngrid = ngrid;
  p->nxrows = nxrows;
  HPX_HIP(hipMemcpyAsync(p->flags, flags, (size_t)nbl * N, hipMemcpyDeviceToDevice, st));
  p->dense_noise = ninv_dense ? (wb ? 2 : 1) : 0;
  hpx_devbuf ones, tmp;                    // dense noise only
  if (ninv_dense) {
    HPX_REQUIRE(nih_dense && (!any_flags || wb),
                "hpx_plan_set_static_dense: needs sqrtm(Ninv) and unflagged data (the reference's column-masked "
                "Ni = Ninv diag(w) is not Hermitian, pspec.py:361: hpx_plan_set_static_dense_flagged)");
    if (wb) {      // flagged channels per baseline (host lists), correction systems, masked residual
      std::vector<uint8_t> hf((size_t)nbl * N);
      HPX_HIP(hipMemcpyAsync(hf.data(), flags, hf.size(), hipMemcpyDeviceToHost, st));
      HPX_HIP(hipStreamSynchronize(st));
      std::vector<int32_t> cnt(nbl, 0);
      int fmax = 0;
      for (int b = 0; b < nbl; ++b) {
        for (int j = 0; j < N; ++j) cnt[b] += hf[(size_t)b * N + j] ? 0 : 1;
        fmax = std::max(fmax, cnt[b]);
      }
      HPX_REQUIRE(T + fmax <= TP, "hpx_plan_set_static_dense_flagged: the plan has too few right-hand-side columns "
                                  "(hpx_plan_create_ex with extra_rhs >= the largest number of flagged channels)");
      HPX_REQUIRE(fmax <= 512, "hpx_plan_set_static_dense_flagged: at most 512 flagged channels per baseline");
      {   // the residual kernel keeps (2 M + N / (16 slices)) x TP doubles in LDS (k_resid): say so HERE, not at the first run
        int P = 4;                       // (the slices of k_resid's launch, post_solve)
        while (P > 1 && (N % P != 0 || N / P < 64)) P >>= 1;
        const size_t lds = (size_t)(2 * M * TP + (N / P) * (TP / 16)) * sizeof(double);
        if (lds > (size_t)160 * 1024) {
          hpx_set_error("hpx_plan_set_static_dense_flagged: %d right-hand-side columns (%d times + %d flagged channels, "
                        "padded) need %zu bytes of LDS in the residual kernel, the CU has 160 KiB: at Nfreqs = %d and "
                        "%d modes at most %d columns", TP, T, fmax, lds, N, M,
                        (int)((160 * 1024 / sizeof(double)) / (2 * M + N / 16.0)) / 16 * 16);
          return HPX_EINVAL;
        }
      }
      p->wb_fmax = fmax > 0 ? fmax : 1;
      std::vector<int32_t> list((size_t)nbl * p->wb_fmax, 0);
      for (int b = 0; b < nbl; ++b) {
        int k = 0;
        for (int j = 0; j < N; ++j)
          if (!hf[(size_t)b * N + j]) list[(size_t)b * p->wb_fmax + k++] = j;
      }
      HPX_TRY(dev_alloc(p, &p->wb_flist, list.size()));
      HPX_TRY(dev_alloc(p, &p->wb_fcount, (size_t)nbl));
      HPX_TRY(dev_alloc(p, &p->wb_W, (size_t)nbl * p->wb_fmax * (p->wb_fmax + T) * 2));
      HPX_TRY(dev_alloc(p, &p->RDre, (size_t)nbl * NP * TP));
      HPX_TRY(dev_alloc(p, &p->RDim, (size_t)nbl * NP * TP));
      HPX_HIP(hipMemcpy(p->wb_flist, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      HPX_HIP(hipMemcpy(p->wb_fcount, cnt.data(), cnt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    const size_t msz = (size_t)nbl * NP * NP;
    HPX_TRY(dev_alloc(p, &p->NIre, msz)); HPX_TRY(dev_alloc(p, &p->NIim, msz));
    HPX_TRY(dev_alloc(p, &p->CDre, msz)); HPX_TRY(dev_alloc(p, &p->CDim, msz));
    hipLaunchKernelGGL(k_dense_planar, dim3(64, nbl), dim3(256), 0, st, ninv_dense, noise_shared, p->NIre, p->NIim, N, NP);
    hipLaunchKernelGGL(k_take_diag, dim3(4, nbl), dim3(256), 0, st, p->NIre, p->ninv, N, NP);   // chi^2 uses Ninv.diagonal()
    HPX_HIP(hipGetLastError());
    HPX_TRY(ones.alloc((size_t)nbl * N));
    std::vector<double> h1((size_t)nbl * N, 1.0);
    HPX_HIP(hipMemcpyAsync(ones.p, h1.data(), h1.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HPX_HIP(hipStreamSynchronize(st));     // h1 goes out of scope with this block's caller frame only at return; be explicit
  } else {
    HPX_HIP(hipMemcpyAsync(p->ninv, ninv, (size_t)nbl * N * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  HPX_HIP(hipMemcpyAsync(p->pmap, prior_map, (size_t)(prior_shared ? 1 : nbl) * N * sizeof(int32_t),
                         hipMemcpyDeviceToDevice, st));
  if (nxrows > 0) {
    HPX_TRY(dev_alloc(p, &p->xgrid, (size_t)nxrows * ngrid));
    HPX_HIP(hipMemcpyAsync(p->xgrid, xgrid, (size_t)nxrows * ngrid * sizeof(double),
                           hipMemcpyDeviceToDevice, st));
  }
  HPX_TRY(hpx_fop_to_planar(fop, p->Fopre, p->Fopim, N, NP, st));
  const double* fgp = fgmodes ? fgmodes : vis;   // never dereferenced when M == 0
  if (M > 0) {
    const long tot = (long)(fg_shared ? 1 : nbl) * N * M;
    hipLaunchKernelGGL(k_fg_planar, dim3(256), dim3(256), 0, st, fgmodes, p->Fre, p->Fim, tot);
    HPX_HIP(hipGetLastError());
  }
  const double isn = 1.0 / sqrt((double)N);
  if (!ninv_dense) {
    hipLaunchKernelGGL(k_prep, dim3(128, nbl), dim3(256), 0, st, vis, flags, ninv, fgp, p->fg_shared,
                       omega, p->Zre, p->Zim, p->Dre, p->Dim, p->ni, T, N, M, NP, TP, MP, p->ncolR, p->omega_mod, 0);
    HPX_HIP(hipGetLastError());
  } else {
    // Z = Ninv [d | F | .] + Ninv^1/2 [omega_b | 0]: the operand block with unit weights (into R as
    // scratch), then two dense products on the MFMA (the stored planar matrices are Hermitian:
    // buffer[k][x] = conj(W[x][k]), hence conjW = 1)
    const long mstr = (long)NP * NP, zstr = (long)NP * p->ncolR;
    hipLaunchKernelGGL(k_prep, dim3(128, nbl), dim3(256), 0, st, vis, flags, ones.p, fgp, p->fg_shared,
                       (const double*)nullptr, p->Rre, p->Rim, p->Dre, p->Dim, p->ni, T, N, M, NP, TP, MP, p->ncolR, 0,
                       wb);
    if (wb) hipLaunchKernelGGL(k_wb_inject, dim3(nbl), dim3(256), 0, st, p->Rre, p->wb_flist, p->wb_fcount,
                               p->wb_fmax, T, NP, p->ncolR);
    HPX_HIP(hipGetLastError());
    HPX_TRY(hpx_launch_dft(nbl, NP, p->ncolR, p->NIre, p->NIim, 1, p->Rre, p->Rim, zstr, p->ncolR, nullptr, 0,
                           p->Zre, p->Zim, zstr, p->ncolR, 1.0, st, 0, mstr));
    hipLaunchKernelGGL(k_take_diag, dim3(4, nbl), dim3(256), 0, st, p->NIre, p->ni, N, NP);
    if (omega) {
      const size_t msz = (size_t)nbl * NP * NP, osz = (size_t)nbl * NP * TP;
      HPX_TRY(tmp.alloc((wb ? 4 : 2) * msz + 4 * osz));
      double *hre = tmp.p, *him = hre + msz, *ore = him + msz, *oim = ore + osz, *ure = oim + osz, *uim = ure + osz;
      if (wb) {     // sqrtm of the column-masked Ni is a general matrix: the product below (conjW = 1) wants its
                    // conjugate transpose stored; one matrix per baseline
        double *gre = uim + osz, *gim = gre + msz;
        hipLaunchKernelGGL(k_dense_planar, dim3(64, nbl), dim3(256), 0, st, nih_dense, 0, gre, gim, N, NP);
        hipLaunchKernelGGL(k_conj_transpose, dim3(64, nbl), dim3(256), 0, st, gre, gim, hre, him, NP);
      } else
      hipLaunchKernelGGL(k_dense_planar, dim3(64, nbl), dim3(256), 0, st, nih_dense, noise_shared, hre, him, N, NP);
      hipLaunchKernelGGL(k_prep_omega_b, dim3(32, nbl), dim3(256), 0, st, omega, ore, oim, T, N, NP, TP, p->omega_mod);
      HPX_HIP(hipGetLastError());
      HPX_TRY(hpx_launch_dft(nbl, NP, TP, hre, him, 1, ore, oim, (long)NP * TP, TP, nullptr, 0, ure, uim,
                             (long)NP * TP, TP, 1.0, st, 0, mstr));
      hipLaunchKernelGGL(k_add_block, dim3(32, nbl), dim3(256), 0, st, p->Zre, p->Zim, zstr, p->ncolR, ure, uim,
                         (long)NP * TP, TP, NP, TP);
      HPX_HIP(hipGetLastError());
    }
    // C = U^H Ninv U = F Ninv F^H / N = F (F Ninv)^H / N  (C is Hermitian): two transforms and a
    // conjugate transpose; CD doubles as scratch for F Ninv
    HPX_TRY(hpx_launch_dft(nbl, NP, NP, p->Fopre, p->Fopim, 0, p->NIre, p->NIim, mstr, NP, nullptr, 0,
                           p->CDre, p->CDim, mstr, NP, 1.0, st, N == NP));
    hpx_devbuf a1h;
    HPX_TRY(a1h.alloc(2 * (size_t)nbl * NP * NP));
    hipLaunchKernelGGL(k_conj_transpose, dim3(64, nbl), dim3(256), 0, st, p->CDre, p->CDim, a1h.p,
                       a1h.p + (size_t)nbl * NP * NP, NP);
    HPX_HIP(hipGetLastError());
    HPX_TRY(hpx_launch_dft(nbl, NP, NP, p->Fopre, p->Fopim, 0, a1h.p, a1h.p + (size_t)nbl * NP * NP, mstr, NP,
                           nullptr, 0, p->CDre, p->CDim, mstr, NP, 1.0 / (double)N, st, N == NP));
    HPX_HIP(hipStreamSynchronize(st));     // a1h is released here
  }
  // R = U^H Z = F Z / sqrt(N)
  HPX_TRY(hpx_launch_dft(nbl, NP, p->ncolR, p->Fopre, p->Fopim, 0, p->Zre, p->Zim,
                         (long)NP * p->ncolR, p->ncolR, nullptr, 0, p->Rre, p->Rim,
                         (long)NP * p->ncolR, p->ncolR, isn, st, N == NP));
  if (!p->dense_noise) {
    hipLaunchKernelGGL(k_circ, dim3(4, nbl), dim3(256), 0, st, p->Rre, p->Rim, p->Cre, p->Cim, N, NP,
                       p->ncolR, TP + MP);
    HPX_HIP(hipGetLastError());
  } else {
    HPX_HIP(hipMemsetAsync(p->Cre, 0, (size_t)nbl * N * sizeof(double), st));
    HPX_HIP(hipMemsetAsync(p->Cim, 0, (size_t)nbl * N * sizeof(double), st));
  }
  if (M > 0) {
    hipLaunchKernelGGL(k_small, dim3(nbl), dim3(256), 0, st, fgp, p->fg_shared, p->Zre, p->Zim,
                       p->Hre, p->Him, p->P4re, p->P4im, N, M, NP, TP, MP, p->ncolR);
    HPX_HIP(hipGetLastError());
  }
  if (omega) {   // P2 = U^H omega_a (shared by all baselines): use G scratch of baseline 0
    hipLaunchKernelGGL(k_prep_omega, dim3(64), dim3(256), 0, st, omega, p->Gre, p->Gim, T, N, NP, TP);
    HPX_HIP(hipGetLastError());
    HPX_TRY(hpx_launch_dft(1, NP, TP, p->Fopre, p->Fopim, 0, p->Gre, p->Gim, 0, TP, nullptr, 0,
                           p->P2re, p->P2im, 0, TP, isn, st, N == NP));
  } else {
    HPX_HIP(hipMemsetAsync(p->P2re, 0, (size_t)NP * TP * sizeof(double), st));
    HPX_HIP(hipMemsetAsync(p->P2im, 0, (size_t)NP * TP * sizeof(double), st));
  }
  // edge tiles for the factor (circulant mode): invariant rows >= rmin of the columns < rmin, P2 by row tile
  p->have_edge = 0;
  {
    const int rmin = 32 * (N / 32);
    if (!p->dense_noise && rmin > 0) {
      p->have_static = 1;                  // (gen_of reads the plan as it stands)
      hpx_gen_batch B = hpx_gen_of(p);
      B.rmin = rmin;
      B.e_bstride = (long)((p->ld - rmin) / 16) * rmin * 32;
      hipLaunchKernelGGL(k_build_edge, dim3(32, nbl), dim3(256), 0, st, B, p->E, p->npad, p->ld);
      hipLaunchKernelGGL(k_p2_tiles, dim3(64), dim3(256), 0, st, p->P2re, p->P2im, p->P2Tre, p->P2Tim, NP, TP);
      HPX_HIP(hipGetLastError());
      p->have_edge = 1;
    }
  }
  HPX_HIP(hipStreamSynchronize(st));
  p->have_static = 1;
  if (p->dense_noise) p->solver = HPX_SOLVER_DENSE;
  return HPX_OK;
}

extern "C" int hpx_plan_set_static(hpx_plan* p, const double* vis, const uint8_t* flags,
                                   const double* ninv, const double* fgmodes, int fg_shared,
                                   const int32_t* prior_map, const double* xgrid, int nxrows,
                                   int prior_shared, int ngrid, const double* omega,
                                   const double* fop, int any_flags, void* stream) {
  HPX_REQUIRE(ninv, "hpx_plan_set_static: null argument");
  return set_static_impl(p, vis, flags, ninv, nullptr, nullptr, 0, fgmodes, fg_shared, prior_map, xgrid, nxrows,
                         prior_shared, ngrid, omega, fop, any_flags, stream);
}

extern "C" int hpx_plan_set_static_dense(hpx_plan* p, const double* vis, const uint8_t* flags,
                                         const double* ninv_dense, const double* nih_dense, int noise_shared,
                                         const double* fgmodes, int fg_shared, const int32_t* prior_map,
                                         const double* xgrid, int nxrows, int prior_shared, int ngrid,
                                         const double* omega, const double* fop, int any_flags, void* stream) {
  HPX_REQUIRE(ninv_dense && nih_dense, "hpx_plan_set_static_dense: null noise matrices");
  return set_static_impl(p, vis, flags, nullptr, ninv_dense, nih_dense, noise_shared, fgmodes, fg_shared, prior_map,
                         xgrid, nxrows, prior_shared, ngrid, omega, fop, any_flags, stream);
}

extern "C" int hpx_plan_set_static_dense_flagged(hpx_plan* p, const double* vis, const uint8_t* flags,
                                                 const double* ninv_dense, int noise_shared,
                                                 const double* nih_masked, const double* fgmodes, int fg_shared,
                                                 const int32_t* prior_map, const double* xgrid, int nxrows,
                                                 int prior_shared, int ngrid, const double* omega, const double* fop,
                                                 void* stream) {
  HPX_REQUIRE(ninv_dense && nih_masked, "hpx_plan_set_static_dense_flagged: null noise matrices");
  return set_static_impl(p, vis, flags, nullptr, ninv_dense, nih_masked, noise_shared, fgmodes, fg_shared, prior_map,
                         xgrid, nxrows, prior_shared, ngrid, omega, fop, 1, stream, 1);
}

static int set_static_impl(hpx_plan* p, const double* vis, const uint8_t* flags,
                           const double* ninv, const double* ninv_dense, const double* nih_dense,
                           int noise_shared, const double* fgmodes, int fg_shared,
                           const int32_t* prior_map, const double* xgrid, int nxrows,
                           int prior_shared, int ngrid, const double* omega,
                           const double* fop, int any_flags, void* stream, int wb);

// `ninv_td` / `nih_td` non-NULL: full noise matrices per (baseline, time) -- the units of the child are then
// dense-noise systems (with the Woodbury correction when a unit has flagged channels), and `ninv_t` is ignored
// (the diagonals of ninv_td take its place for chi^2 and the parent's surrogate)
static int pertime_impl(hpx_plan* p, const double* vis, const uint8_t* flags_t, const double* ninv_t,
                        const double* ninv_td, const double* nih_td, const double* fgmodes, int fg_shared,
                        const int32_t* prior_map, const double* xgrid, int nxrows, int prior_shared, int ngrid,
                        const double* omega, const double* fop, int any_flags, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int nbl = p->nbl, T = p->T, N = p->N, M = p->M, NP = p->NP, TP = p->TP;
  hpx_devbuf dgb;
  int wb = 0, fmax = 0;
  if (ninv_td) {
    HPX_TRY(dgb.alloc((size_t)nbl * T * N));
    hipLaunchKernelGGL(k_pt_diag, dim3(4, nbl * T), dim3(256), 0, st, ninv_td, dgb.p, N);
    HPX_HIP(hipGetLastError());
    ninv_t = dgb.p;
    std::vector<uint8_t> hf((size_t)nbl * T * N);
    HPX_HIP(hipMemcpyAsync(hf.data(), flags_t, hf.size(), hipMemcpyDeviceToHost, st));
    HPX_HIP(hipStreamSynchronize(st));
    for (size_t u = 0; u < (size_t)nbl * T; ++u) {
      int f = 0;
      for (int j = 0; j < N; ++j) f += hf[u * N + j] ? 0 : 1;
      fmax = std::max(fmax, f);
    }
    wb = fmax > 0;
  }
  // 1. the parent's time-independent parts (foreground planes, operator, prior tables, omega_a block)
  //    with the any-time mask -- its own solve operators are never used in this mode
  hpx_devbuf tmp;
  HPX_TRY(tmp.alloc((size_t)nbl * N + ((size_t)nbl * N + 7) / 8 + 8));
  double* nany = tmp.p;
  uint8_t* fany = (uint8_t*)(tmp.p + (size_t)nbl * N);
  hipLaunchKernelGGL(k_pt_reduce, dim3(4, nbl), dim3(256), 0, st, flags_t, ninv_t, fany, nany, T, N);
  HPX_HIP(hipGetLastError());
  HPX_TRY(set_static_impl(p, vis, fany, nany, nullptr, nullptr, 0, fgmodes, fg_shared, prior_map, xgrid, nxrows,
                          prior_shared, ngrid, omega, fop, any_flags, stream));
  // 2. per-time data, flags, noise
  HPX_TRY(dev_alloc(p, &p->flags_t, (size_t)nbl * T * N));
  HPX_TRY(dev_alloc(p, &p->ninv_t, (size_t)nbl * T * N));
  HPX_HIP(hipMemcpyAsync(p->flags_t, flags_t, (size_t)nbl * T * N, hipMemcpyDeviceToDevice, st));
  HPX_HIP(hipMemcpyAsync(p->ninv_t, ninv_t, (size_t)nbl * T * N * sizeof(double), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_pt_data, dim3(64, nbl), dim3(256), 0, st, vis, flags_t, p->Dre, p->Dim, T, N, NP, TP);
  HPX_HIP(hipGetLastError());
  // 3. the child: nbl*T units of one time sample each
  if (p->child) { hpx_plan_destroy(p->child); p->child = nullptr; }
  HPX_TRY(hpx_plan_create_impl(&p->child, nbl * T, 1, N, M, wb ? fmax : 0));
  hpx_plan* c = p->child;
  c->omega_mod = T;
  hpx_devbuf fgx;
  const double* fgc = fgmodes;
  if (M > 0 && !fg_shared) {
    const long per = (long)N * M * 2;
    HPX_TRY(fgx.alloc((size_t)nbl * T * per));
    hipLaunchKernelGGL(k_pt_expand_fg, dim3(16, nbl * T), dim3(256), 0, st, fgmodes, fgx.p, T, per);
    HPX_HIP(hipGetLastError());
    fgc = fgx.p;
  }
  // (vis (nbl,T,N) is (nbl*T,1,N); flags_t / ninv_t (nbl,T,N) are (nbl*T,N); the child never draws: no priors)
  if (ninv_td) {
    HPX_TRY(set_static_impl(c, vis, flags_t, nullptr, ninv_td, nih_td, 0, fgc, fg_shared, p->pmap, nullptr, 0, 1,
                            ngrid, omega, fop, wb, stream, wb));
    HPX_TRY(dev_alloc(p, &p->RDre, (size_t)nbl * NP * TP));     // masked residual for the quadratic form
    HPX_TRY(dev_alloc(p, &p->RDim, (size_t)nbl * NP * TP));
  } else
  HPX_TRY(set_static_impl(c, vis, flags_t, ninv_t, nullptr, nullptr, 0, fgc, fg_shared, p->pmap, nullptr, 0, 1,
                          ngrid, omega, fop, any_flags, stream));
  HPX_TRY(dev_alloc(p, &p->PTre, (size_t)T * NP * c->TP));
  HPX_TRY(dev_alloc(p, &p->PTim, (size_t)T * NP * c->TP));
  hipLaunchKernelGGL(k_pt_p2, dim3(64), dim3(256), 0, st, p->P2re, p->P2im, p->PTre, p->PTim, T, NP, TP, c->TP);
  HPX_HIP(hipGetLastError());
  // ... and by row tile for the factor's edge tiles: unit u = (b, t) has one right-hand-side row, time t
  HPX_TRY(dev_alloc(p, &p->PTTre, (size_t)T * NP * 16));
  HPX_TRY(dev_alloc(p, &p->PTTim, (size_t)T * NP * 16));
  HPX_HIP(hipMemsetAsync(p->PTTre, 0, (size_t)T * NP * 16 * sizeof(double), st));
  HPX_HIP(hipMemsetAsync(p->PTTim, 0, (size_t)T * NP * 16 * sizeof(double), st));
  hipLaunchKernelGGL(k_pt_p2t, dim3(64), dim3(256), 0, st, p->P2re, p->P2im, p->PTTre, p->PTTim, T, NP, TP);
  HPX_HIP(hipGetLastError());
  HPX_HIP(hipStreamSynchronize(st));
  p->per_time = ninv_td ? 2 : 1;
  p->solver = HPX_SOLVER_DENSE;
  return HPX_OK;
}

extern "C" int hpx_plan_set_static_pertime(hpx_plan* p, const double* vis, const uint8_t* flags_t,
                                           const double* ninv_t, const double* fgmodes, int fg_shared,
                                           const int32_t* prior_map, const double* xgrid, int nxrows,
                                           int prior_shared, int ngrid, const double* omega,
                                           const double* fop, int any_flags, void* stream) {
  HPX_REQUIRE(p && vis && flags_t && ninv_t && fop && prior_map, "hpx_plan_set_static_pertime: null argument");
  return pertime_impl(p, vis, flags_t, ninv_t, nullptr, nullptr, fgmodes, fg_shared, prior_map, xgrid, nxrows,
                      prior_shared, ngrid, omega, fop, any_flags, stream);
}

extern "C" int hpx_plan_set_static_pertime_dense(hpx_plan* p, const double* vis, const uint8_t* flags_t,
                                                 const double* ninv_t_dense, const double* nih_t,
                                                 const double* fgmodes, int fg_shared, const int32_t* prior_map,
                                                 const double* xgrid, int nxrows, int prior_shared, int ngrid,
                                                 const double* omega, const double* fop, int any_flags,
                                                 void* stream) {
  HPX_REQUIRE(p && vis && flags_t && ninv_t_dense && nih_t && fop && prior_map,
              "hpx_plan_set_static_pertime_dense: null argument");
  return pertime_impl(p, vis, flags_t, nullptr, ninv_t_dense, nih_t, fgmodes, fg_shared, prior_map, xgrid, nxrows,
                      prior_shared, ngrid, omega, fop, any_flags, stream);
}
