"""tangram_amd -- MI355X-native drop-in for the hot path of broadinstitute/Tangram:
`map_cells_to_space` -> `Mapper` / `MapperConstrained` training loop, running in hand-written HIP kernels
behind a C ABI (include/tangram_hip.h).  Importing this package does not import scanpy."""
from .mapping_optimizer import Mapper, MapperConstrained          # noqa: F401
from .mapping_utils import map_cells_to_space, adata_to_cluster_expression, density_priors  # noqa: F401
from . import preprocess                                          # noqa: F401
from .utils import project_genes                                  # noqa: F401
from .batched import train_many, MapperBatch                                   # noqa: F401
from .cross_validation import cross_val, cv_data_gen                        # noqa: F401

__version__ = "0.1.0"
