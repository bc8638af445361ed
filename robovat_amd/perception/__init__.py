"""Host-side camera model and point-cloud grouping with the reference's API
(``robovat/perception``): used by tests as the reference-shaped pipeline the
device point clouds are checked against, and by callers that want the
projection math on the host.  The product observation path is on the device
(``csrc/rv_dev_obs.h``)."""
from robovat_amd.perception.camera import Camera, intrinsic_to_projection_matrix  # noqa: F401
from robovat_amd.perception import point_cloud_utils  # noqa: F401
