"""Host-side runtime pieces around the HIP path (no kernels here)."""
from runtime.graphed_step import FlatGrads, GraphedTrainStep, batch_signature  # noqa: F401
from runtime.prefetch import GeometryPrefetcher  # noqa: F401
from runtime.gc_schedule import ScheduledGC  # noqa: F401
