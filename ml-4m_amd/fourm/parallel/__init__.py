from .data_parallel import DataParallel, GradReducer, cap_collective_channels, DEFAULT_RESERVED_CUS
