// gfbe_gnss.h — the GNSS factors of the window, __host__ __device__ arithmetic:
//   GnssPsrDoppFactor::Evaluate   Ground-Fusion++/vins_estimator/src/factor/gnss_psr_dopp_factor.cpp:50-208
//   DtDdtFactor, DdtSmoothFactor  factor/gnss_dt_ddt_factor.cpp, factor/gnss_ddt_smooth_factor.cpp
// and the gnss_comm functions the pseudo-range factor calls on every evaluation (the package IS in the reference tree):
//   ecef2geo / geo2rotation / ecef2rotation / sat_azel    gnss_comm/src/gnss_utility.cpp:347-383, 745-772
//   calculate_trop_delay (Saastamoinen + Niell mapping)   gnss_comm/src/gnss_utility.cpp:774-862
//   calculate_ion_delay (Klobuchar)                       gnss_comm/src/gnss_utility.cpp:865-899
// What the factor's CONSTRUCTOR derives from the observation and the ephemeris (:3-47: satellite position / velocity / clock at
// the transmission time, group delay, the two URA scalings) crosses the ABI precomputed in gfbe_gnss_obs: ephemeris propagation
// is front-end work, the per-evaluation arithmetic is what sits on the optimisation path.
#pragma once
#include "gfbe_math.h"
#include "../../include/gfbe.h"

namespace gfd {

#define GNSS_LIGHT_SPEED 2.99792458e8        // gnss_constant.hpp:214
#define GNSS_EARTH_OMG 7.2921151467e-5       // :208
#define GNSS_ECCE_2 6.69437999014e-3         // :203
#define GNSS_SEMI_MAJOR 6378137.0            // :205
#define GNSS_PI 3.14159265358979323846
#define GNSS_PSR_TO_DOPP_RATIO 5.0           // gnss_psr_dopp_factor.hpp:11

// latitude / longitude in DEGREES, height in metres (gnss_utility.cpp:347-383)
GF_HD vec3 gnss_ecef2geo(const vec3 &xyz) {
  if (xyz[0] == 0.0 && xyz[1] == 0.0) return mk3(0.0, 0.0, 0.0);
  const double e2 = GNSS_ECCE_2, a = GNSS_SEMI_MAJOR, a2 = a * a, b2 = a2 * (1.0 - e2), b = sqrt(b2), ep2 = (a2 - b2) / b2;
  const double p = sqrt(xyz[0] * xyz[0] + xyz[1] * xyz[1]);
  double s1 = xyz[2] * a, s2 = p * b, h = sqrt(s1 * s1 + s2 * s2);
  const double sin_theta = s1 / h, cos_theta = s2 / h;
  s1 = xyz[2] + ep2 * b * sin_theta * sin_theta * sin_theta;
  s2 = p - a * e2 * cos_theta * cos_theta * cos_theta;
  h = sqrt(s1 * s1 + s2 * s2);
  const double sin_lat = s1 / h, cos_lat = s2 / h;
  const double lat = atan(s1 / s2);
  const double N = a2 / sqrt(a2 * cos_lat * cos_lat + b2 * sin_lat * sin_lat);
  return mk3(lat * (180.0 / GNSS_PI), atan2(xyz[1], xyz[0]) * (180.0 / GNSS_PI), p / cos_lat - N);
}
// R_ecef_enu of a geodetic position (gnss_utility.cpp:745-755)
GF_HD mat3 gnss_geo2rotation(const vec3 &lla) {
  const double lat = lla[0] * (GNSS_PI / 180.0), lon = lla[1] * (GNSS_PI / 180.0);
  const double sl = sin(lat), cl = cos(lat), so = sin(lon), co = cos(lon);
  mat3 R;
  R(0, 0) = -so; R(0, 1) = -sl * co; R(0, 2) = cl * co;
  R(1, 0) = co;  R(1, 1) = -sl * so; R(1, 2) = cl * so;
  R(2, 0) = 0.0; R(2, 1) = cl;       R(2, 2) = sl;
  return R;
}
// azimuth / elevation of the satellite seen from the receiver (gnss_utility.cpp:762-772)
GF_HD void gnss_sat_azel(const vec3 &rcv, const vec3 &sat, const vec3 &rcv_lla, double *azel) {
  const vec3 d = sub(sat, rcv);
  const double n = sqrt(dot3(d, d));
  const vec3 u = mk3(d[0] / n, d[1] / n, d[2] / n);
  const vec3 enu = tmv(gnss_geo2rotation(rcv_lla), u);       // ecef2enu = R_ecef_enu^T
  azel[0] = sqrt(u[0] * u[0] + u[1] * u[1]) < 1e-12 ? 0.0 : atan2(enu[0], enu[1]);
  if (azel[0] < 0.0) azel[0] += 2.0 * GNSS_PI;
  azel[1] = asin(enu[2]);
}
GF_HD double gnss_interpc(const double *coef, double lat) {
  const int i = (int)(lat / 15.0);
  if (i < 1) return coef[0];
  if (i > 4) return coef[4];
  return coef[i - 1] * (1.0 - lat / 15.0 + i) + coef[i] * (lat / 15.0 - i);
}
GF_HD double gnss_mapf(double el, double a, double b, double c) {
  const double s = sin(el);
  return (1.0 + a / (1.0 + b / (1.0 + c))) / (s + (a / (s + b / (s + c))));
}
// Saastamoinen zenith delays on the standard atmosphere x Niell mapping functions (gnss_utility.cpp:774-862); doy = time2doy(t)
GF_HD double gnss_trop_delay(double doy, const vec3 &lla, const double *azel) {
  if (lla[2] < -100.0 || 1e4 < lla[2] || azel[1] <= 0.0) return 0.0;
  const double hgt = lla[2] < 0.0 ? 0.0 : lla[2];
  const double pres = 1013.25 * pow(1.0 - 2.2557e-5 * hgt, 5.2568);
  const double temp = 15.0 - 6.5e-3 * hgt + 273.16;
  const double e = 6.108 * 0.7 * exp((17.15 * temp - 4684.0) / (temp - 38.45));
  const double zhd = 0.0022768 * pres / (1.0 - 0.00266 * cos(2.0 * lla[0] * (GNSS_PI / 180.0)) - 0.00028 * hgt / 1e3);
  const double zwd = 0.002277 * (1255.0 / temp + 0.05) * e;
  const double coef[9][5] = {
      {1.2769934e-3, 1.2683230e-3, 1.2465397e-3, 1.2196049e-3, 1.2045996e-3}, {2.9153695e-3, 2.9152299e-3, 2.9288445e-3, 2.9022565e-3, 2.9024912e-3},
      {62.610505e-3, 62.837393e-3, 63.721774e-3, 63.824265e-3, 64.258455e-3}, {0.0, 1.2709626e-5, 2.6523662e-5, 3.4000452e-5, 4.1202191e-5},
      {0.0, 2.1414979e-5, 3.0160779e-5, 7.2562722e-5, 11.723375e-5},          {0.0, 9.0128400e-5, 4.3497037e-5, 84.795348e-5, 170.37206e-5},
      {5.8021897e-4, 5.6794847e-4, 5.8118019e-4, 5.9727542e-4, 6.1641693e-4}, {1.4275268e-3, 1.5138625e-3, 1.4572752e-3, 1.5007428e-3, 1.7599082e-3},
      {4.3472961e-2, 4.6729510e-2, 4.3908931e-2, 4.4626982e-2, 5.4736038e-2}};
  const double el = azel[1];
  double lat = lla[0];
  const double y = (doy - 28.0) / 365.25 + (lat < 0.0 ? 0.5 : 0.0);
  const double cosy = cos(2.0 * GNSS_PI * y);
  lat = fabs(lat);
  double ah[3], aw[3];
  for (int i = 0; i < 3; i++) { ah[i] = gnss_interpc(coef[i], lat) - gnss_interpc(coef[i + 3], lat) * cosy; aw[i] = gnss_interpc(coef[i + 6], lat); }
  const double dm = (1.0 / sin(el) - gnss_mapf(el, 2.53e-5, 5.49e-3, 1.14e-3)) * lla[2] / 1e3;
  const double mapfw = gnss_mapf(el, aw[0], aw[1], aw[2]), mapfh = gnss_mapf(el, ah[0], ah[1], ah[2]) + dm;
  return mapfh * zhd + mapfw * zwd;
}
// Klobuchar broadcast model (gnss_utility.cpp:865-899); ion = 8 parameters or null; tow = time2gpst(t) (seconds of the GPS week)
GF_HD double gnss_ion_delay(double tow, const double *ion, const vec3 &lla, const double *azel) {
  if (!ion) return 0.0;
  if (lla[2] < -1e3 || azel[1] <= 0.0) return 0.0;
  const double psi = 0.0137 / (azel[1] / GNSS_PI + 0.11) - 0.022;
  double phi = lla[0] / 180.0 + psi * cos(azel[0]);
  if (phi > 0.416) phi = 0.416; else if (phi < -0.416) phi = -0.416;
  const double lam = lla[1] / 180.0 + psi * sin(azel[0]) / cos(phi * GNSS_PI);
  phi += 0.064 * cos((lam - 1.617) * GNSS_PI);
  double tt = 43200.0 * lam + tow;
  tt -= floor(tt / 86400.0) * 86400.0;
  const double f0 = 0.53 - azel[1] / GNSS_PI, f = 1.0 + 16.0 * f0 * f0 * f0;
  double amp = ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3]));
  double per = ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7]));
  amp = amp < 0.0 ? 0.0 : amp;
  per = per < 72000.0 ? 72000.0 : per;
  const double x = 2.0 * GNSS_PI * (tt - 50400.0) / per;
  return GNSS_LIGHT_SPEED * f * (fabs(x) < 1.57 ? 5e-9 + amp * (1.0 + x * x * (-0.5 + x * x / 24.0)) : 5e-9);
}

// GnssPsrDoppFactor::Evaluate (gnss_psr_dopp_factor.cpp:50-208). r[2]; J[2][18] (may be null): columns P_i(3) V_i(3) P_j(3) V_j(3)
// rcv_dt rcv_ddt yaw_enu_local anc_ecef(3) — the non-zero parts of the reference's 2 x {7, 9, 7, 9, 1, 1, 1, 3} blocks. As in the
// reference the Jacobian leaves out the derivatives of the atmosphere and Sagnac terms and takes d/d anc_ecef "for simplicity".
GF_HD void gnss_psr_dopp_eval(const gfbe_gnss_obs &o, const double *ion, const double *Pi, const double *Vi, const double *Pj, const double *Vj,
                              double rcv_dt, double rcv_ddt, double yaw, const double *anc, double *r, double *J) {
  const double ratio = o.ratio;
  const vec3 lp = mk3(ratio * Pi[0] + (1.0 - ratio) * Pj[0], ratio * Pi[1] + (1.0 - ratio) * Pj[1], ratio * Pi[2] + (1.0 - ratio) * Pj[2]);
  const vec3 lv = mk3(ratio * Vi[0] + (1.0 - ratio) * Vj[0], ratio * Vi[1] + (1.0 - ratio) * Vj[1], ratio * Vi[2] + (1.0 - ratio) * Vj[2]);
  const double sy = sin(yaw), cy = cos(yaw);
  mat3 Rel;       // R_enu_local
  Rel(0, 0) = cy; Rel(0, 1) = -sy; Rel(0, 2) = 0.0; Rel(1, 0) = sy; Rel(1, 1) = cy; Rel(1, 2) = 0.0; Rel(2, 0) = 0.0; Rel(2, 1) = 0.0; Rel(2, 2) = 1.0;
  const vec3 ref = ld3(anc);
  const mat3 Ree = gnss_geo2rotation(gnss_ecef2geo(ref)), R = mul(Ree, Rel);      // R_ecef_enu, R_ecef_local
  const vec3 P = add(mv(R, lp), ref), V = mv(R, lv);
  const vec3 sp = ld3(o.sv_pos), sv = ld3(o.sv_vel);
  double ion_delay = 0.0, tro_delay = 0.0, azel[2] = {0.0, GNSS_PI / 2.0};
  if (dot3(P, P) > 0.0) {
    const vec3 lla = gnss_ecef2geo(P);
    gnss_sat_azel(P, sp, lla, azel);
    tro_delay = gnss_trop_delay(o.doy, lla, azel);
    ion_delay = gnss_ion_delay(o.tow, ion, lla, azel);
  }
  const double sin_el = sin(azel[1]), sin_el_2 = sin_el * sin_el;
  const double pr_weight = sin_el_2 / o.pr_uura * 10.0;                            // relative_sqrt_info = 10 (:46)
  const double dp_weight = sin_el_2 / o.dp_uura * 10.0 * GNSS_PSR_TO_DOPP_RATIO;
  const vec3 d = sub(sp, P);
  const double range = sqrt(dot3(d, d));
  const vec3 u = mk3(d[0] / range, d[1] / range, d[2] / range);
  const double psr_sagnac = GNSS_EARTH_OMG * (sp[0] * P[1] - sp[1] * P[0]) / GNSS_LIGHT_SPEED;
  const double psr_est = range + psr_sagnac + rcv_dt - o.svdt * GNSS_LIGHT_SPEED + ion_delay + tro_delay + o.tgd * GNSS_LIGHT_SPEED;
  r[0] = (psr_est - o.psr) * pr_weight;
  const double dopp_sagnac = GNSS_EARTH_OMG / GNSS_LIGHT_SPEED * (sv[0] * P[1] + sp[0] * V[1] - sv[1] * P[0] - sp[1] * V[0]);
  const vec3 dv = sub(sv, V);
  const double dopp_est = dot3(dv, u) + dopp_sagnac + rcv_ddt - o.svddt * GNSS_LIGHT_SPEED;
  r[1] = (dopp_est + o.dopp * o.wavelength) * dp_weight;
  if (!J) return;
  for (int q = 0; q < 36; q++) J[q] = 0.0;
  const vec3 uR = tmv(R, u);                       // (rcv2sat_unit^T R_ecef_local)^T
  // unit2rcv_pos = -(|d|^2 I - d d^T) / |d|^3; row vector (sv_vel - V)^T unit2rcv_pos R
  const double n2 = range * range, n3 = n2 * range;
  vec3 w;
  for (int j = 0; j < 3; j++) {
    double s = 0.0;
    for (int i = 0; i < 3; i++) s += dv[i] * -(((i == j) ? n2 : 0.0) - d[i] * d[j]) / n3;
    w[j] = s;
  }
  const vec3 wR = tmv(R, w);
  for (int j = 0; j < 3; j++) {
    J[0 + j] = -uR[j] * pr_weight * ratio;          J[18 + j] = wR[j] * dp_weight * ratio;                    // P_i
    J[18 + 3 + j] = -uR[j] * dp_weight * ratio;                                                               // V_i (Doppler row only)
    J[6 + j] = -uR[j] * pr_weight * (1.0 - ratio);  J[18 + 6 + j] = wR[j] * dp_weight * (1.0 - ratio);        // P_j
    J[18 + 9 + j] = -uR[j] * dp_weight * (1.0 - ratio);                                                       // V_j
    J[15 + j] = -u[j] * pr_weight;                                                                            // anc_ecef (pseudo-range row)
  }
  J[12] = pr_weight;                                // rcv_dt: pseudo-range row
  J[18 + 13] = dp_weight;                           // rcv_ddt: Doppler row
  {                                                 // yaw_enu_local: -u . (R_ecef_enu dRz/dyaw x)
    const vec3 dp = mv(Ree, mk3(-sy * lp[0] - cy * lp[1], cy * lp[0] - sy * lp[1], 0.0));
    const vec3 dvv = mv(Ree, mk3(-sy * lv[0] - cy * lv[1], cy * lv[0] - sy * lv[1], 0.0));
    J[14] = -dot3(u, dp) * pr_weight;
    J[18 + 14] = -dot3(u, dvv) * dp_weight;
  }
}

// DtDdtFactor (gnss_dt_ddt_factor.cpp:3-34, dt_info_coeff = 50) and DdtSmoothFactor (gnss_ddt_smooth_factor.cpp:3-22): linear, their
// Jacobians are the constants {-50, 50, -25 dt, -25 dt} and {w, -w}
GF_HD double gnss_dt_ddt_res(double dt_i, double dt_j, double ddt_i, double ddt_j, double delta_t) {
  return (dt_j - dt_i - 0.5 * (ddt_i + ddt_j) * delta_t) * 50.0;
}
GF_HD double gnss_ddt_smooth_res(double ddt_i, double ddt_j, double weight) { return (ddt_i - ddt_j) * weight; }

}  // namespace gfd
