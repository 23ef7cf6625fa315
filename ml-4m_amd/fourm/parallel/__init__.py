from .data_parallel import DataParallel, GradReducer
