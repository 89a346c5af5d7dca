package com.android.nQuant;

import java.io.IOException;

/* CIELAB variant: same constructor signature as the reference (NQ/PnnLABQuantizer.java:24); convert() is inherited. */
public class PnnLABQuantizer extends PnnQuantizer {
	public PnnLABQuantizer(String fname) throws IOException {
		super(fname);
	}

	@Override
	protected int kind() { return 1; }
}
