"""CPU oracle for the video-saliency hot path -- TEST INFRASTRUCTURE ONLY.

A torch-CPU fp32 restatement of the reference algorithms on the path named by
BASELINE.json (`video_features_pytorch/mask.py`, `grad_cam_videos.py`,
`models/I3D_doubled{,_kth}.py`, `models/convolution_lstm.py`, `models/CLSTM_4.py`
and the search loop of `FindMasksComparison_I3D_smth.py:176-251`).  Every
function cites the reference file:line it follows.

Pinned: `tests/golden/*.npz` hold outputs of the REFERENCE modules themselves,
captured in the build container by `tests/golden/make_golden.py` (which imports
/root/reference); `tests/test_oracle_vs_golden.py` checks this restatement
against them.  The one step that is NOT pinned is `cv2.resize` inside Grad-CAM
(OpenCV is not installed here; reference call site grad_cam_videos.py:119):
`gradcam_ref.resize_bilinear` restates OpenCV's INTER_LINEAR rule and the golden
vectors for that step were produced with this same rule -- "parity unpinned"
for the resize step only.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this package.  The product (`interpreting-video-features_amd/`) never
does: it calls the HIP library and fails loudly when that is missing.
"""
