// oracle/gfo_gnss.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle). PARITY UNPINNED against the reference binary (gnss_comm needs
// ROS + glog to build); pinned by the independent numpy restatement, the central-difference checks and the closed-form cases of
// tests/test_gnss_oracle.py (zenith satellite, equator / pole geodesy, Klobuchar night-time floor).
// The GNSS factors of the window, written the way the reference's Evaluate reads, with Eigen
// expressions replaced by gfo_math.h operators:
//   GnssPsrDoppFactor::Evaluate   Ground-Fusion++/vins_estimator/src/factor/gnss_psr_dopp_factor.cpp:50-208
//   DtDdtFactor::Evaluate         factor/gnss_dt_ddt_factor.cpp:3-34
//   DdtSmoothFactor::Evaluate     factor/gnss_ddt_smooth_factor.cpp:3-22
//   ecef2geo / ecef2enu / geo2rotation / sat_azel / calculate_trop_delay / calculate_ion_delay
//                                 gnss_comm/src/gnss_utility.cpp:347-383, 730-772, 774-862, 865-899
#include <cmath>
#include <cstring>

#include "gfo_api.h"
#include "gfo_math.h"

using namespace gfo;

namespace {

const double kC = 2.99792458e8, kOmega = 7.2921151467e-5, kE2 = 6.69437999014e-3, kA = 6378137.0, kD2R = M_PI / 180.0, kR2D = 180.0 / M_PI;

V3 ecef2geo(V3 xyz) {      // gnss_utility.cpp:347-383
  if (xyz.x == 0 && xyz.y == 0) return v3(0, 0, 0);
  const double a2 = kA * kA, b2 = a2 * (1 - kE2), b = std::sqrt(b2), ep2 = (a2 - b2) / b2, p = std::sqrt(xyz.x * xyz.x + xyz.y * xyz.y);
  double s1 = xyz.z * kA, s2 = p * b, h = std::sqrt(s1 * s1 + s2 * s2);
  const double sin_theta = s1 / h, cos_theta = s2 / h;
  s1 = xyz.z + ep2 * b * std::pow(sin_theta, 3);
  s2 = p - kA * kE2 * std::pow(cos_theta, 3);
  h = std::sqrt(s1 * s1 + s2 * s2);
  const double tan_lat = s1 / s2, sin_lat = s1 / h, cos_lat = s2 / h;
  const double N = a2 * std::pow(a2 * cos_lat * cos_lat + b2 * sin_lat * sin_lat, -0.5);
  return v3(std::atan(tan_lat) * kR2D, std::atan2(xyz.y, xyz.x) * kR2D, p / cos_lat - N);
}
M3 geo2rotation(V3 geo) {   // R_ecef_enu, :745-755
  const double lat = geo.x * kD2R, lon = geo.y * kD2R, sl = std::sin(lat), cl = std::cos(lat), so = std::sin(lon), co = std::cos(lon);
  M3 R = {{{-so, -sl * co, cl * co}, {co, -sl * so, cl * so}, {0, cl, sl}}};
  return R;
}
V3 ecef2enu(V3 geo, V3 v) {  // :730-743
  const double lat = geo.x * kD2R, lon = geo.y * kD2R, sl = std::sin(lat), cl = std::cos(lat), so = std::sin(lon), co = std::cos(lon);
  M3 R = {{{-so, co, 0}, {-sl * co, -sl * so, cl}, {cl * co, cl * so, sl}}};
  return R * v;
}
void sat_azel(V3 rcv, V3 sat, double *azel) {   // :762-772
  const V3 lla = ecef2geo(rcv);
  const V3 d = sat - rcv, u = (1.0 / norm(d)) * d, enu = ecef2enu(lla, u);
  azel[0] = std::sqrt(u.x * u.x + u.y * u.y) < 1e-12 ? 0.0 : std::atan2(enu.x, enu.y);
  azel[0] += azel[0] < 0 ? 2 * M_PI : 0;
  azel[1] = std::asin(enu.z);
}
double interpc(const double *coef, double lat) {   // :775-781
  const int i = (int)(lat / 15.0);
  if (i < 1) return coef[0];
  if (i > 4) return coef[4];
  return coef[i - 1] * (1.0 - lat / 15.0 + i) + coef[i] * (lat / 15.0 - i);
}
double mapf(double el, double a, double b, double c) {   // :783-787
  const double sinel = std::sin(el);
  return (1.0 + a / (1.0 + b / (1.0 + c))) / (sinel + (a / (sinel + b / (sinel + c))));
}
double nmf(double doy, V3 pos, const double *azel, double *mapfw) {   // :789-839 (pos in degrees)
  static const double coef[][5] = {
      {1.2769934E-3, 1.2683230E-3, 1.2465397E-3, 1.2196049E-3, 1.2045996E-3}, {2.9153695E-3, 2.9152299E-3, 2.9288445E-3, 2.9022565E-3, 2.9024912E-3},
      {62.610505E-3, 62.837393E-3, 63.721774E-3, 63.824265E-3, 64.258455E-3}, {0.0000000E-0, 1.2709626E-5, 2.6523662E-5, 3.4000452E-5, 4.1202191E-5},
      {0.0000000E-0, 2.1414979E-5, 3.0160779E-5, 7.2562722E-5, 11.723375E-5}, {0.0000000E-0, 9.0128400E-5, 4.3497037E-5, 84.795348E-5, 170.37206E-5},
      {5.8021897E-4, 5.6794847E-4, 5.8118019E-4, 5.9727542E-4, 6.1641693E-4}, {1.4275268E-3, 1.5138625E-3, 1.4572752E-3, 1.5007428E-3, 1.7599082E-3},
      {4.3472961E-2, 4.6729510E-2, 4.3908931E-2, 4.4626982E-2, 5.4736038E-2}};
  const double aht[] = {2.53E-5, 5.49E-3, 1.14E-3};
  double ah[3], aw[3], el = azel[1], lat = pos.x, hgt = pos.z;
  if (el <= 0.0) { if (mapfw) *mapfw = 0.0; return 0.0; }
  const double y = (doy - 28.0) / 365.25 + (lat < 0.0 ? 0.5 : 0.0), cosy = std::cos(2.0 * M_PI * y);
  lat = std::fabs(lat);
  for (int i = 0; i < 3; i++) { ah[i] = interpc(coef[i], lat) - interpc(coef[i + 3], lat) * cosy; aw[i] = interpc(coef[i + 6], lat); }
  const double dm = (1.0 / std::sin(el) - mapf(el, aht[0], aht[1], aht[2])) * hgt / 1E3;
  if (mapfw) *mapfw = mapf(el, aw[0], aw[1], aw[2]);
  return mapf(el, ah[0], ah[1], ah[2]) + dm;
}
double trop_delay(double doy, V3 lla, const double *azel) {   // :841-862
  const double temp0 = 15.0, humi = 0.7;
  if (lla.z < -100.0 || 1E4 < lla.z || azel[1] <= 0) return 0.0;
  const double hgt = lla.z < 0.0 ? 0.0 : lla.z;
  const double pres = 1013.25 * std::pow(1.0 - 2.2557E-5 * hgt, 5.2568), temp = temp0 - 6.5E-3 * hgt + 273.16;
  const double e = 6.108 * humi * std::exp((17.15 * temp - 4684.0) / (temp - 38.45));
  const double zhd = 0.0022768 * pres / (1.0 - 0.00266 * std::cos(2.0 * lla.x * kD2R) - 0.00028 * hgt / 1E3);
  const double zwd = 0.002277 * (1255.0 / temp + 0.05) * e;
  double mapfw, mapfh = nmf(doy, lla, azel, &mapfw);
  return mapfh * zhd + mapfw * zwd;
}
double ion_delay(double tow, const double *ion, V3 lla, const double *azel) {   // :865-899
  if (lla.z < -1E3 || azel[1] <= 0) return 0.0;
  double psi = 0.0137 / (azel[1] / M_PI + 0.11) - 0.022;
  double phi = lla.x / 180.0 + psi * std::cos(azel[0]);
  if (phi > 0.416) phi = 0.416; else if (phi < -0.416) phi = -0.416;
  const double lam = lla.y / 180.0 + psi * std::sin(azel[0]) / std::cos(phi * M_PI);
  phi += 0.064 * std::cos((lam - 1.617) * M_PI);
  double tt = 43200.0 * lam + tow;
  tt -= std::floor(tt / 86400.0) * 86400.0;
  const double f = 1.0 + 16.0 * std::pow(0.53 - azel[1] / M_PI, 3.0);
  double amp = ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3])), per = ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7]));
  amp = amp < 0.0 ? 0.0 : amp;
  per = per < 72000.0 ? 72000.0 : per;
  const double x = 2.0 * M_PI * (tt - 50400.0) / per;
  return kC * f * (std::fabs(x) < 1.57 ? 5E-9 + amp * (1.0 + x * x * (-0.5 + x * x / 24.0)) : 5E-9);
}

}  // namespace

// GnssPsrDoppFactor::Evaluate (gnss_psr_dopp_factor.cpp:50-208) for one observation. r[2]; J[2][18] or null, columns
// P_i(3) V_i(3) P_j(3) V_j(3) rcv_dt rcv_ddt yaw_enu_local anc_ecef(3): the non-zero parts of the reference's 2 x {7, 9, 7, 9, 1, 1, 1, 3}.
namespace gfo {
void eval_gnss_psr_dopp(const gfbe_gnss_obs &o, const double *iono, const double *pi, const double *vi, const double *pj, const double *vj,
                        double rcv_dt, double rcv_ddt, double yaw, const double *anc, double *r, double *J) {
  const V3 Pi = v3(pi), Vi = v3(vi), Pj = v3(pj), Vj = v3(vj);
  const double ratio = o.ratio;
  const V3 ref = v3(anc), sv_pos = v3(o.sv_pos), sv_vel = v3(o.sv_vel);
  const V3 local_pos = ratio * Pi + (1.0 - ratio) * Pj, local_vel = ratio * Vi + (1.0 - ratio) * Vj;
  const double sy = std::sin(yaw), cy = std::cos(yaw);
  const M3 R_enu_local = {{{cy, -sy, 0}, {sy, cy, 0}, {0, 0, 1}}};
  const M3 R_ecef_enu = geo2rotation(ecef2geo(ref)), R_ecef_local = R_ecef_enu * R_enu_local;
  const V3 P_ecef = R_ecef_local * local_pos + ref, V_ecef = R_ecef_local * local_vel;
  double ion = 0, tro = 0, azel[2] = {0, M_PI / 2.0};
  if (norm(P_ecef) > 0) {
    sat_azel(P_ecef, sv_pos, azel);
    const V3 lla = ecef2geo(P_ecef);
    tro = trop_delay(o.doy, lla, azel);
    ion = iono ? ion_delay(o.tow, iono, lla, azel) : 0.0;
  }
  const double sin_el = std::sin(azel[1]), sin_el_2 = sin_el * sin_el;
  const double pr_weight = sin_el_2 / o.pr_uura * 10.0, dp_weight = sin_el_2 / o.dp_uura * 10.0 * 5.0;
  const V3 rcv2sat = sv_pos - P_ecef, unit = (1.0 / norm(rcv2sat)) * rcv2sat;
  const double psr_sagnac = kOmega * (sv_pos.x * P_ecef.y - sv_pos.y * P_ecef.x) / kC;
  const double psr_est = norm(rcv2sat) + psr_sagnac + rcv_dt - o.svdt * kC + ion + tro + o.tgd * kC;
  const double dopp_sagnac = kOmega / kC * (sv_vel.x * P_ecef.y + sv_pos.x * V_ecef.y - sv_vel.y * P_ecef.x - sv_pos.y * V_ecef.x);
  const double dopp_est = dot(sv_vel - V_ecef, unit) + dopp_sagnac + rcv_ddt - o.svddt * kC;
  const double r0 = (psr_est - o.psr) * pr_weight, r1 = (dopp_est + o.dopp * o.wavelength) * dp_weight;
  r[0] = r0; r[1] = r1;
  if (!J) return;
  std::memset(J, 0, sizeof(double) * 36);
  const double n2 = norm2(rcv2sat), n3 = std::pow(norm(rcv2sat), 3);
  M3 unit2rcv;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) unit2rcv.m[i][j] = -(i == j ? (n2 - get(rcv2sat, i) * get(rcv2sat, i)) / n3 : (-get(rcv2sat, i) * get(rcv2sat, j)) / n3);
  const V3 uR = T(R_ecef_local) * unit;                              // (unit^T R)^T
  const V3 wR = T(R_ecef_local) * (T(unit2rcv) * (sv_vel - V_ecef));  // ((sv_vel - V)^T unit2rcv R)^T
  for (int j = 0; j < 3; j++) {
    J[j] = -get(uR, j) * pr_weight * ratio;            J[18 + j] = get(wR, j) * dp_weight * ratio;
    J[18 + 3 + j] = -get(uR, j) * dp_weight * ratio;
    J[6 + j] = -get(uR, j) * pr_weight * (1.0 - ratio); J[18 + 6 + j] = get(wR, j) * dp_weight * (1.0 - ratio);
    J[18 + 9 + j] = -get(uR, j) * dp_weight * (1.0 - ratio);
    J[15 + j] = -get(unit, j) * pr_weight;
  }
  J[12] = pr_weight;
  J[18 + 13] = dp_weight;
  const M3 d_yaw = {{{-sy, -cy, 0}, {cy, -sy, 0}, {0, 0, 0}}};
  J[14] = -dot(unit, R_ecef_enu * (d_yaw * local_pos)) * pr_weight;
  J[18 + 14] = -dot(unit, R_ecef_enu * (d_yaw * local_vel)) * dp_weight;
}
}  // namespace gfo

extern "C" int32_t gfo_gnss_eval(void *, int32_t n_obs, const gfbe_gnss_obs *obs, const double *iono, const gfbe_state *st, const gfbe_gnss_state *g,
                                 const double *frame_dt, double ddt_weight, double *r_obs, double *J_obs, double *r_dt_ddt, double *r_smooth,
                                 double *cost) {
  const int W = GFBE_WINDOW_SIZE;
  double c = 0.0;
  for (int k = 0; k < n_obs; k++) {
    const gfbe_gnss_obs &o = obs[k];
    double r[2];
    gfo::eval_gnss_psr_dopp(o, iono, st->para_Pose[o.lower_idx], st->para_SpeedBias[o.lower_idx], st->para_Pose[o.lower_idx + 1],
                            st->para_SpeedBias[o.lower_idx + 1], g->rcv_dt[o.frame][o.sys_idx], g->rcv_ddt[o.frame], g->yaw_enu_local,
                            g->anc_ecef, r, J_obs ? J_obs + (size_t)36 * k : nullptr);
    c += 0.5 * r[0] * r[0] + 0.5 * r[1] * r[1];
    if (r_obs) { r_obs[2 * k] = r[0]; r_obs[2 * k + 1] = r[1]; }
  }
  for (int sys = 0; sys < 4; sys++)
    for (int i = 0; i < W; i++) {
      const double avg = 0.5 * (g->rcv_ddt[i] + g->rcv_ddt[i + 1]);
      const double r = (g->rcv_dt[i + 1][sys] - g->rcv_dt[i][sys] - avg * frame_dt[i]) * 50.0;
      c += 0.5 * r * r;
      if (r_dt_ddt) r_dt_ddt[sys * W + i] = r;
    }
  for (int i = 0; i < W; i++) {
    const double r = (g->rcv_ddt[i] - g->rcv_ddt[i + 1]) * ddt_weight;
    c += 0.5 * r * r;
    if (r_smooth) r_smooth[i] = r;
  }
  if (cost) *cost = c;
  return GFBE_OK;
}
