// av1_ops_cdfs.hpp — the default CDF image of a tile in the slot layout of av1_ops.hpp, built from the specification's default
// tables (host/av1_default_cdfs.inc).  Host code only (the GPU coder uploads the image once per context).
#pragma once
#include <vector>
#include "av1_ops.hpp"
#include "av1_ops32.hpp"
#include "../host/av1_default_cdfs.inc"

namespace av1ops {

// cdfs[t.off[slot] ..]: nsym - 1 inverse values, then the counter (0), then padding
inline std::vector<uint16_t> default_slot_image(bool key, int qcat, SlotTable *t) {
  build_slot_table(key, t);
  std::vector<uint16_t> img((size_t)t->words, 0);
  auto put = [&](int slot, const uint16_t *spec, int nsym) {
    for (int i = 0; i < nsym - 1; i++) img[t->off[slot] + i] = (uint16_t)(32768 - spec[i]);
  };
  for (int i = 0; i < 3; i++) put(S_SKIP + i, Default_Skip_Cdf[i], 2);
  put(S_PART8, Default_Partition_W8_Cdf[0], 4);
  for (int i = 0; i < 4; i++) { put(S_PART16 + i, Default_Partition_W16_Cdf[i], 10); put(S_PART32 + i, Default_Partition_W32_Cdf[i], 10); put(S_PART64 + i, Default_Partition_W64_Cdf[i], 10); }
  put(S_USE_WIENER, Default_Use_Wiener_Cdf[0], 2);
  put(S_TXB_SKIP_Y, Default_Txb_Skip_Cdf[qcat][1][0], 2);
  for (int i = 0; i < 3; i++) put(S_TXB_SKIP_C + i, Default_Txb_Skip_Cdf[qcat][0][7 + i], 2);
  put(S_EOB64_Y, Default_Eob_Pt_64_Cdf[qcat][0][0], 7);
  put(S_EOB16_C, Default_Eob_Pt_16_Cdf[qcat][1][0], 5);
  for (int i = 0; i < 5; i++) put(S_EOBX_Y + i, Default_Eob_Extra_Cdf[qcat][1][0][i], 2);
  for (int i = 0; i < 3; i++) put(S_EOBX_C + i, Default_Eob_Extra_Cdf[qcat][0][1][i], 2);
  for (int i = 0; i < 3; i++) { put(S_DC_SIGN_Y + i, Default_Dc_Sign_Cdf[qcat][0][i], 2); put(S_DC_SIGN_C + i, Default_Dc_Sign_Cdf[qcat][1][i], 2); }
  for (int i = 0; i < 4; i++) { put(S_BASE_EOB_Y + i, Default_Coeff_Base_Eob_Cdf[qcat][1][0][i], 3); put(S_BASE_EOB_C + i, Default_Coeff_Base_Eob_Cdf[qcat][0][1][i], 3); }
  for (int i = 0; i < 26; i++) { put(S_BASE_Y + i, Default_Coeff_Base_Cdf[qcat][1][0][i], 4); put(S_BASE_C + i, Default_Coeff_Base_Cdf[qcat][0][1][i], 4); }
  for (int i = 0; i < 21; i++) { put(S_BR_Y + i, Default_Coeff_Br_Cdf[qcat][1][0][i], 4); put(S_BR_C + i, Default_Coeff_Br_Cdf[qcat][0][1][i], 4); }
  if (key) {
    for (int a = 0; a < 5; a++) for (int l = 0; l < 5; l++) put(S_KF_Y_MODE + a * 5 + l, Default_Intra_Frame_Y_Mode_Cdf[a][l], 13);
    for (int m = 0; m < 13; m++) { put(S_UV_MODE + m, Default_Uv_Mode_Cfl_Allowed_Cdf[m], 14); put(S_INTRA_TX + m, Default_Intra_Tx_Type_Set1_Cdf[1][m], 7); }
    for (int i = 0; i < 8; i++) put(S_ANGLE + i, Default_Angle_Delta_Cdf[i], 7);
  } else {
    for (int i = 0; i < 4; i++) put(S_IS_INTER + i, Default_Is_Inter_Cdf[i], 2);
    static const int kBit[3] = { 0, 2, 3 };      // single_ref_p1, p3, p4
    for (int c = 0; c < 2; c++) for (int k = 0; k < 3; k++) put(S_SINGLE_REF + c * 3 + k, Default_Single_Ref_Cdf[1 + c][kBit[k]], 2);
    for (int i = 0; i < 6; i++) { put(S_NEW_MV + i, Default_New_Mv_Cdf[i], 2); put(S_REF_MV + i, Default_Ref_Mv_Cdf[i], 2); }
    put(S_ZERO_MV, Default_Zero_Mv_Cdf[0], 2);
    for (int i = 0; i < 3; i++) put(S_DRL + i, Default_Drl_Mode_Cdf[i], 2);
    put(S_MV_JOINT, Default_Mv_Joint_Cdf[0], 4);
    for (int c = 0; c < 2; c++) {
      const int b = S_MV_COMP + c * 16;
      put(b + MVC_CLASS, Default_Mv_Class_Cdf[0], 11);
      put(b + MVC_CLASS0, Default_Mv_Class0_Bit_Cdf[0], 2);
      put(b + MVC_CLASS0_FR, Default_Mv_Class0_Fr_Cdf[0], 4);
      put(b + MVC_CLASS0_FR + 1, Default_Mv_Class0_Fr_Cdf[1], 4);
      put(b + MVC_SIGN, Default_Mv_Sign_Cdf[0], 2);
      for (int i = 0; i < 10; i++) put(b + MVC_BITS + i, Default_Mv_Bit_Cdf[i], 2);
      put(b + MVC_FR, Default_Mv_Fr_Cdf[0], 4);
    }
    put(S_INTER_TX, Default_Inter_Tx_Type_Set1_Cdf[1], 16);
  }
  return img;
}

// the same for the tiles of a key frame's 32x32 band (av1_ops32.hpp): luma 32x32 (transform-size context 3, plane type 0), chroma
// 16x16 (context 2, plane type 1)
inline std::vector<uint16_t> default_slot_image_k32(int qcat, SlotTable *t) {
  build_slot_table_k32(t);
  std::vector<uint16_t> img((size_t)t->words, 0);
  auto put = [&](int slot, const uint16_t *spec, int nsym) {
    for (int i = 0; i < nsym - 1; i++) img[t->off[slot] + i] = (uint16_t)(32768 - spec[i]);
  };
  put(K_SKIP, Default_Skip_Cdf[0], 2);
  put(K_PART32, Default_Partition_W32_Cdf[0], 10); put(K_PART64, Default_Partition_W64_Cdf[0], 10);
  put(K_USE_WIENER, Default_Use_Wiener_Cdf[0], 2);
  put(K_TXB_SKIP_Y, Default_Txb_Skip_Cdf[qcat][3][0], 2);
  for (int i = 0; i < 3; i++) put(K_TXB_SKIP_C + i, Default_Txb_Skip_Cdf[qcat][2][7 + i], 2);
  put(K_EOB_Y, Default_Eob_Pt_1024_Cdf[qcat][0][0], 11);
  put(K_EOB_C, Default_Eob_Pt_256_Cdf[qcat][1][0], 9);
  for (int i = 0; i < 9; i++) put(K_EOBX_Y + i, Default_Eob_Extra_Cdf[qcat][3][0][i], 2);
  for (int i = 0; i < 7; i++) put(K_EOBX_C + i, Default_Eob_Extra_Cdf[qcat][2][1][i], 2);
  for (int i = 0; i < 3; i++) { put(K_DC_SIGN_Y + i, Default_Dc_Sign_Cdf[qcat][0][i], 2); put(K_DC_SIGN_C + i, Default_Dc_Sign_Cdf[qcat][1][i], 2); }
  for (int i = 0; i < 4; i++) { put(K_BASE_EOB_Y + i, Default_Coeff_Base_Eob_Cdf[qcat][3][0][i], 3); put(K_BASE_EOB_C + i, Default_Coeff_Base_Eob_Cdf[qcat][2][1][i], 3); }
  for (int i = 0; i < 26; i++) { put(K_BASE_Y + i, Default_Coeff_Base_Cdf[qcat][3][0][i], 4); put(K_BASE_C + i, Default_Coeff_Base_Cdf[qcat][2][1][i], 4); }
  for (int i = 0; i < 21; i++) { put(K_BR_Y + i, Default_Coeff_Br_Cdf[qcat][3][0][i], 4); put(K_BR_C + i, Default_Coeff_Br_Cdf[qcat][2][1][i], 4); }
  for (int a = 0; a < 5; a++) for (int l = 0; l < 5; l++) put(K_KF_Y_MODE + a * 5 + l, Default_Intra_Frame_Y_Mode_Cdf[a][l], 13);
  for (int m = 0; m < 13; m++) put(K_UV_MODE + m, Default_Uv_Mode_Cfl_Allowed_Cdf[m], 14);
  for (int i = 0; i < 8; i++) put(K_ANGLE + i, Default_Angle_Delta_Cdf[i], 7);
  return img;
}

}  // namespace av1ops
