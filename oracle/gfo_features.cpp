// oracle/gfo_features.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
// Landmark bookkeeping, restated from
//   FeatureManager::getFeatureCount   VE/estimator/feature_manager.cpp:43-55
//   FeatureManager::getDepthVector    VE/estimator/feature_manager.cpp:286-302
//   FeatureManager::setDepth          VE/estimator/feature_manager.cpp:249-267
//   visual-factor loops               VE/estimator/estimator.cpp:3326-3358 and 3498-3531
// Integer outputs must be bit-exact against the product's gfbe_build_visual_factors.
#include "gfo_api.h"

extern "C" {

int32_t gfo_feature_count(const gfbe_feature_list *fl) {
  int32_t cnt = 0;
  for (int f = 0; f < fl->n; f++)
    if (fl->n_obs[f] >= 4) cnt++;
  return cnt;
}

int32_t gfo_visual_factor_count(const gfbe_feature_list *fl, int32_t only0) {
  int32_t k = 0;
  for (int f = 0; f < fl->n; f++) {
    if (fl->n_obs[f] < 4) continue;
    if (only0 && fl->start_frame[f] != 0) continue;
    k += fl->n_obs[f] - 1;
  }
  return k;
}

int32_t gfo_build_visual_factors(const gfbe_feature_list *fl, int32_t only0, int32_t *feature_index, int32_t *imu_i,
                                 int32_t *imu_j, double *pts_i, double *pts_j, double *vel_i, double *vel_j,
                                 double *td_i, double *td_j, double *para_Feature, uint8_t *feature_const) {
  int32_t k = 0, index = -1;
  for (int f = 0; f < fl->n; f++) {
    if (fl->n_obs[f] < 4) continue;
    ++index;
    if (para_Feature) para_Feature[index] = 1.0 / fl->estimated_depth[f];
    if (feature_const) feature_const[index] = fl->estimate_flag[f] == 1 ? 1 : 0;
    const int i = fl->start_frame[f];
    if (only0 && i != 0) continue;
    const double *first = fl->obs + 7 * (size_t)fl->obs_offset[f];
    int j = i - 1;
    for (int o = 0; o < fl->n_obs[f]; o++) {
      j++;
      if (i == j) continue;
      const double *row = fl->obs + 7 * (size_t)(fl->obs_offset[f] + o);
      feature_index[k] = index; imu_i[k] = i; imu_j[k] = j;
      for (int c = 0; c < 3; c++) { pts_i[3 * k + c] = first[c]; pts_j[3 * k + c] = row[c]; }
      for (int c = 0; c < 2; c++) { vel_i[2 * k + c] = first[5 + c]; vel_j[2 * k + c] = row[5 + c]; }
      td_i[k] = fl->obs_td[fl->obs_offset[f]];
      td_j[k] = fl->obs_td[fl->obs_offset[f] + o];
      k++;
    }
  }
  return k;
}

void gfo_set_depth(const gfbe_feature_list *fl, const double *para_Feature, double *estimated_depth, int32_t *solve_flag) {
  int index = -1;
  for (int f = 0; f < fl->n; f++) {
    if (fl->n_obs[f] < 4) continue;
    estimated_depth[f] = 1.0 / para_Feature[++index];
    solve_flag[f] = estimated_depth[f] < 0 ? 2 : 1;
  }
}

}  // extern "C"
