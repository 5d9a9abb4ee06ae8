from .caspr_dataset import DynamicPCLDataset, load_seq_path, load_time_data, parse_dataset_cfg, select_item  # noqa: F401
