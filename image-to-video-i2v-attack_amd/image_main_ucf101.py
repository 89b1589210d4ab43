"""Drop-in for `/root/reference/image_main_ucf101.py`: `image_main.py` with the UCF-101 twin's differences (`--step` defaults to 10
and reaches the ensemble class, `video_names = str(val_label)`, no adaptive-ensemble branch) and the UCF-101 loader in front of it
(`/root/reference/dataset_ucf101.py`):

  * `--frame_dir ROOT --setting test01_setting.txt --used_idxs used_idxs.pkl`: the reference's own inputs -- jpg frame folders
    `ROOT/<video_path>/image_{:05d}.jpg` (:36-45), the clip list (:81-99) reduced to the index list (:59-64), `LoopPadding(32)`
    (transforms_ucf101.py:23-40) -- read by `i2v_amd.clips.ucf101_batches` (Pillow decoding on the `--workers` threads, pinned
    host batches one ahead of the GPU);
  * or `--clip_dir` of `{label}-raw.npy` already-decoded uint8 clips;

either way the loader's validation transform runs on the device as ONE kernel, bit-exact against Pillow (`:113-126`: PIL BILINEAR
`Scale(224)` -> `CornerCrop(224, 'c')` -> `ToTensor` -> `Normalize`; `Engine.clip_resample_crop`)."""
import image_main


def main(argv=None):
    return image_main.main(argv, ucf101=True)


if __name__ == "__main__":
    main()
