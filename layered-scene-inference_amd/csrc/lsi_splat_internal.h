// Internal (non-ABI) declarations shared by the splat translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/lsi_hip.h"

struct SplatArgs {
  LsiSplatDesc d;
  const float* tex;
  const float* disp;
  const float* mask;
  const float* M;
  float* out_img;
  float* out_wts;
  float* out_disp;
  float* canvas;  // caller's workspace: ATOMIC canvases / STREAM exchange area
  size_t ws_bytes;
  int nch;        // canvas channels (4, or 5 with disparity)
  int ncanv;      // canvases per batch element (1 or L)
  int shared;     // 1: compose without disparity, all layers share a canvas
  int band_rows;  // ROWBAND: target rows per workgroup
  // lsi_splat_fwd_both: the composed outputs next to the per-layer ones (NULL
  // otherwise)
  float* out_img_c;
  float* out_wts_c;
};

// LSI_PATH_STREAM launcher and workspace need (lsi_splat_stream.hip).
size_t lsi_stream_workspace_bytes(const LsiSplatDesc* d);
int lsi_stream_launch(const SplatArgs& a, hipStream_t stream);
// The compact STREAM instance (lsi_splat_stream2.hip): compose mode, no mask,
// unit normaliser, channels-last textures, rows of whole 256-pixel segments.
bool lsi_stream2_applies(const SplatArgs& a, bool simple, int layout);
int lsi_stream2_launch(const SplatArgs& a, int wmax, hipStream_t stream);

// LSI_PATH_TILE launcher and workspace need (lsi_splat_tile.hip).
size_t lsi_tile_workspace_bytes(const LsiSplatDesc* d);
int lsi_tile_launch(const SplatArgs& a, hipStream_t stream);

// The any-pose sweep kernel (lsi_splat_sweep.hip), launched by lsi_tile_launch
// after the disparity ranges (range[(l * B + b) * LSI_RANGE_SLICES + k] =
// {min, max} of a slice of rows) are on the stream.
#define LSI_RANGE_SLICES 8
#define LSI_SWEEP_MAXL 16
int lsi_sweep_launch(const SplatArgs& a, const float2* range, hipStream_t stream);
