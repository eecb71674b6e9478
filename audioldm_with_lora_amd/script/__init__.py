"""Entry-point mirrors of the reference's script/train and script/inference."""
