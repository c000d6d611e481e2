from .base import S3Projection
from .qubit_tapering import QubitTapering
