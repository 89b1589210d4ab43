"""Drop-in for `/root/reference/image_main_ucf101.py`: `image_main.py` with the UCF-101 twin's differences (`--step` defaults to 10
and reaches the ensemble class, `video_names = str(val_label)`, no adaptive-ensemble branch) and, for decoded uint8 clips
(`--clip_dir` of `{label}-raw.npy`, 240 x 320 UCF-101 frames), the UCF-101 loader's validation transform on the device
(`dataset_ucf101.py:113-126`: PIL BILINEAR `Scale(224)` -> `CornerCrop(224, 'c')` -> `ToTensor` -> `Normalize`, one kernel, bit-exact
against Pillow).  Reading the jpg frame folders themselves (`dataset_ucf101.py:14-45`) is decoding and stays outside."""
import image_main


def main(argv=None):
    return image_main.main(argv, ucf101=True)


if __name__ == "__main__":
    main()
