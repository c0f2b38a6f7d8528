"""`indextts` console entry (/root/reference/indextts/cli.py:7-70): text + voice prompt -> wav file."""
import argparse
import os
import sys


def main():
    p = argparse.ArgumentParser(description="IndexTTS Command Line (MI355X HIP engine)")
    p.add_argument("text", type=str, help="Text to be synthesized")
    p.add_argument("-v", "--voice", type=str, required=True, help="Path to the audio prompt file (wav format)")
    p.add_argument("-o", "--output_path", type=str, default="gen.wav")
    p.add_argument("-c", "--config", type=str, default="checkpoints/config.yaml")
    p.add_argument("--model_dir", type=str, default="checkpoints")
    p.add_argument("--fp16", action="store_true", default=True, help="bf16 throughput engine (default)")
    p.add_argument("--fp32", action="store_true", help="fp32 parity engine")
    p.add_argument("-f", "--force", action="store_true", default=False)
    p.add_argument("-d", "--device", type=str, default=None)
    a = p.parse_args()
    if not a.text.strip():
        print("ERROR: Text is empty.")
        p.print_help()
        sys.exit(1)
    if not os.path.exists(a.voice):
        print(f"Audio prompt file {a.voice} does not exist.")
        sys.exit(1)
    if not os.path.exists(a.config):
        print(f"Config file {a.config} does not exist.")
        sys.exit(1)
    if os.path.exists(a.output_path) and not a.force:
        print(f"ERROR: Output file {a.output_path} already exists. Use --force to overwrite.")
        sys.exit(1)
    from indextts.infer import IndexTTS

    tts = IndexTTS(cfg_path=a.config, model_dir=a.model_dir, is_fp16=not a.fp32, device=a.device)
    tts.infer(audio_prompt=a.voice, text=a.text.strip(), output_path=a.output_path)


if __name__ == "__main__":
    main()
