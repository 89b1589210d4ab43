"""Drop-in for `/root/reference/reference_ucf101.py` (the evaluator `run_image_guided.py:22-29` starts for Table 4): `reference.py`
with the UCF-101 twin's differences -- 101 classes (`:125`) and the fine-tuned checkpoints' directory (`:24-31`)."""
import reference


def main(argv=None):
    return reference.main(argv, ucf101=True)


if __name__ == "__main__":
    main()
