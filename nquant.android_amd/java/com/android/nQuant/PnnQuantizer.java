package com.android.nQuant;

/* Host class of the MI355X build: same public surface as the reference's PnnQuantizer (constructor from a file name,
 * Bitmap convert(int nMaxColors, boolean dither) throws Exception, boolean hasAlpha()); the quantization itself runs in
 * libnquant_hip.so through the JNI shim jni/nquant_jni.c.  Not compiled in the build image (no JDK there). */

import android.graphics.Bitmap;
import android.graphics.BitmapFactory;

public class PnnQuantizer {
	static { System.loadLibrary("nquant_jni"); }

	public static final int MODE_REFERENCE_SEQUENTIAL = 0, MODE_PARALLEL_TILED = 1, MODE_LOOKUP_ONLY = 2;

	protected int width, height;
	protected int[] pixels = null;
	protected long handle = 0;
	protected int mode = MODE_PARALLEL_TILED;
	protected long seed = System.nanoTime();     // the reference draws from an unseeded java.util.Random
	protected int[] palette = null;

	protected int kind() { return 0; }           // 0 = PnnQuantizer, 1 = PnnLABQuantizer

	private static native long nqCreate(int kind, int device);
	private static native void nqDestroy(long h);
	private static native int[] nqConvert(long h, int[] argb, int w, int hgt, int nMaxColors, boolean dither, long seed, int mode,
	                                      int[] outArgb, short[] outIndex);
	private static native boolean nqHasAlpha(long h);

	public PnnQuantizer(String fname) {
		Bitmap bitmap = BitmapFactory.decodeFile(fname);
		width = bitmap.getWidth();
		height = bitmap.getHeight();
		pixels = new int[width * height];
		bitmap.getPixels(pixels, 0, width, 0, 0, width, height);
	}

	public void setSeed(long seed) { this.seed = seed; }
	public void setMode(int mode) { this.mode = mode; }
	public int[] getPalette() { return palette; }

	public Bitmap convert(int nMaxColors, boolean dither) throws Exception {
		if (handle == 0)
			handle = nqCreate(kind(), 0);
		int[] qPixels = new int[pixels.length];
		palette = nqConvert(handle, pixels, width, height, nMaxColors, dither, seed, mode, qPixels, null);
		return Bitmap.createBitmap(qPixels, width, height, Bitmap.Config.ARGB_8888);
	}

	public boolean hasAlpha() {
		return handle != 0 && nqHasAlpha(handle);
	}

	/** convert() of many quantizer objects in one call (nq_convert_batch): the merge loops of all images run side by side on
	 *  the GPU.  in[i] / out[i] are DIRECT buffers of width*height ints (page-lock them for fully overlapped copies);
	 *  returns the palettes. Results equal quantizers[i].convert(nMaxColors, dither) with the same seed. */
	public static int[][] convertBatch(PnnQuantizer[] quantizers, java.nio.IntBuffer[] in, java.nio.IntBuffer[] out,
			int nMaxColors, boolean dither) {
		long[] handles = new long[quantizers.length], seeds = new long[quantizers.length];
		int[] widths = new int[quantizers.length], heights = new int[quantizers.length];
		for (int i = 0; i < quantizers.length; ++i) {
			PnnQuantizer q = quantizers[i];
			if (q.handle == 0)
				q.handle = nqCreate(q.kind(), 0);
			handles[i] = q.handle; seeds[i] = q.seed; widths[i] = q.width; heights[i] = q.height;
		}
		int[][] palettes = nqConvertBatch(handles, in, widths, heights, nMaxColors, dither, seeds, quantizers[0].mode, out);
		for (int i = 0; i < quantizers.length; ++i)
			quantizers[i].palette = palettes[i];
		return palettes;
	}
	private static native int[][] nqConvertBatch(long[] handles, java.nio.IntBuffer[] in, int[] widths, int[] heights,
			int nMaxColors, boolean dither, long[] seeds, int mode, java.nio.IntBuffer[] out);

	@Override
	protected void finalize() throws Throwable {
		if (handle != 0) { nqDestroy(handle); handle = 0; }
		super.finalize();
	}
}
